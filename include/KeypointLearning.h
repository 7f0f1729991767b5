// KeypointLearning.h -- drop-in pcl::keypoints::KeypointLearningDetector on top of libkpl.
//
// Same class name, namespace, template parameters, constructor defaults and public methods as
// /root/reference/include/KeypointLearning.h:55-206; the bodies marshal to the C-ABI of
// include/kpl.h, whose HIP kernels run the feature -> forest -> NMS path on an MI355X.
// Header-only: a TestDetector written against the reference header compiles against this one
// (src/main_test_detector.cpp:123-187 uses setNAnnulus/setNBins/setNonMaxima/setNonMaxRadius/
// setNonMaximaDrawsRemove/setPredictionThreshold/setRadiusSearch/loadForest/setInputCloud/
// setNormals/compute).
//
//   -DKPL_USE_PCL      derive from the real pcl::Keypoint (PCL >= 1.8 headers required)
//   -DKPL_USE_OPENCV   computePointsForTrainingFeatures returns cv::Mat (CV_32F) like the reference
// Without them a minimal in-repo shim (kpl_pcl_shim.h) and kpl::FeatureMatrix are used.
//
// Differences to the reference, all documented in DESIGN.md:
//   * without setNormals the normals are estimated on the device the way the reference's initCompute
//     does (impl/KeypointLearning.hpp:125-148): pcl::NormalEstimation with the feature radius for an
//     unorganized cloud (kpl_estimate_normals), pcl::IntegralImageNormalEstimation (SIMPLE_3D_GRADIENT,
//     smoothing 5) for an organized one (kpl_estimate_normals_organized); both flip towards the cloud's
//     sensor origin, like PCL's estimators do by default;
//   * no search tree is ever built on the host: the engine has its own spatial index on the device.  A tree
//     handed to the inherited setSearchMethod is only asked whether it returns SORTED results
//     (pcl::search::KdTree's constructor default!): if so the feature loop meets the neighbors in FLANN's
//     sorted order, ascending (squared distance, index) -- kpl_params::neighbor_order -- otherwise in the
//     engine's canonical order; setSortedSearch(bool) overrides the tree;
//   * only radius search with search surface == input is supported (the reference's k-search mode
//     divides by a zero support, hpp:345; a separate surface mixes index spaces, hpp:332 vs :149);
//   * points with non-finite xyz or normal get score NaN and keep their input index (the
//     reference compacts them away and then indexes with tree indices, hpp:277 vs :213);
//   * n_annulus * n_bins must equal the forest's var_count (checked; OpenCV would read past the
//     sample).
#pragma once

#include <cmath>
#include <iostream>
#include <string>
#include <vector>

#include "kpl.h"

#ifdef KPL_USE_PCL
#include <pcl/keypoints/keypoint.h>
#include <pcl/point_types.h>
#else
#include "kpl_pcl_shim.h"
#endif
#ifdef KPL_USE_OPENCV
#include <opencv2/core.hpp>
#endif

namespace kpl {

// What computePointsForTrainingFeatures returns when OpenCV is not around: M x F row-major
// float matrix with the cv::Mat accessors the reference's caller uses
// (/root/reference/src/main_train_detector.cpp:439-446).
struct FeatureMatrix {
    int rows = 0, cols = 0;
    std::vector<float> data;
    template <typename T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data.data() + (size_t)r * cols); }
    template <typename T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data.data() + (size_t)r * cols); }
    template <typename T> T &at(int r, int c) { return reinterpret_cast<T &>(data[(size_t)r * cols + c]); }
    bool empty() const { return rows == 0; }
};

#ifdef KPL_USE_OPENCV
typedef cv::Mat FeatureMat;
#else
typedef FeatureMatrix FeatureMat;
#endif

}  // namespace kpl

namespace pcl {
namespace keypoints {

template <typename PointInT, typename PointOutT, typename NormalT = pcl::Normal>
class KeypointLearningDetector : public Keypoint<PointInT, PointOutT> {
public:
#ifdef KPL_USE_PCL
    typedef boost::shared_ptr<KeypointLearningDetector<PointInT, PointOutT, NormalT>> Ptr;
    typedef boost::shared_ptr<const KeypointLearningDetector<PointInT, PointOutT, NormalT>> ConstPtr;
#else
    typedef std::shared_ptr<KeypointLearningDetector<PointInT, PointOutT, NormalT>> Ptr;
    typedef std::shared_ptr<const KeypointLearningDetector<PointInT, PointOutT, NormalT>> ConstPtr;
#endif
    typedef typename Keypoint<PointInT, PointOutT>::PointCloudIn PointCloudIn;
    typedef typename Keypoint<PointInT, PointOutT>::PointCloudOut PointCloudOut;
    typedef typename Keypoint<PointInT, PointOutT>::KdTree KdTree;           // reference KeypointLearning.h:66
    typedef typename PointCloudIn::ConstPtr PointCloudInConstPtr;
    typedef pcl::PointCloud<NormalT> PointCloudN;
    typedef typename PointCloudN::Ptr PointCloudNPtr;
    typedef typename PointCloudN::ConstPtr PointCloudNConstPtr;

    // KeypointLearning.h:81 of the reference: same defaults.  `device` = HIP device ordinal.
    KeypointLearningDetector(double prediction_th = 0.5f, bool non_maxima = true,
                             bool non_maxima_draws_remove = true, double non_max_radius = 0.0f,
                             int n_annulus = 5, int n_bins = 10, int device = 0)
        : non_maxima_(non_maxima), non_maxima_draws_remove_(non_maxima_draws_remove),
          non_maxima_draws_threshold_(0.0f), prediction_th_(prediction_th),
          non_maxima_radius_(non_max_radius), n_annulus_(n_annulus), n_bins_(n_bins) {
        this->name_ = "Keypoint_Learnining_Detector";
        create_status_ = kpl_create(&handle_, device);
    }

    virtual ~KeypointLearningDetector() { kpl_destroy(handle_); }
    KeypointLearningDetector(const KeypointLearningDetector &) = delete;
    KeypointLearningDetector &operator=(const KeypointLearningDetector &) = delete;

    virtual void setInputCloud(const PointCloudInConstPtr &cloud) {        // hpp:49-57
        if (normals_ && this->input_ && (cloud != this->input_)) {
            normals_.reset();
            staged_normals_ = false;
        }
        this->input_ = cloud;
        // the view's size is known from here on: the engine's tables for it are sized now, not inside compute() (kpl_reserve)
#ifndef KPL_NO_RESERVE
        if (handle_ && cloud) kpl_reserve(handle_, (int)cloud->points.size(), sizeof(PointInT), sizeof(NormalT));
#endif
        if (host_staging_) stage_points();
    }
    virtual void setNormals(const PointCloudNConstPtr &normals) {
        normals_ = normals;
        if (host_staging_) stage_normals();
    }

    // Opt-in (no reference counterpart): copy the cloud and the normals into pinned staging buffers of the engine
    // WHEN THEY ARE SET (packed 12-byte xyz / normals), so that compute() uploads them by DMA, half the bytes of the
    // PCL records, the normals overlapped with the first index kernels (kpl_host_staging /
    // kpl_detect_keypoints_staged).  The price: setInputCloud / setNormals become O(n) copies, and points or normals
    // modified in place afterwards are not seen until they are set again -- which is why it is off by default (the
    // reference's setters only keep a pointer).
    void setHostStaging(bool on) {
        host_staging_ = on;
        staged_points_ = staged_normals_ = false;
        if (on && this->input_) stage_points();
        if (on && normals_) stage_normals();
    }
    virtual void setNonMaxima(bool non_maxima) { non_maxima_ = non_maxima; }
    virtual void setNonMaximaDrawsRemove(bool v) { non_maxima_draws_remove_ = v; }
    virtual void setNonMaximaDrawsThreshold(float v) { non_maxima_draws_threshold_ = v; }
    virtual void setPredictionThreshold(double th) { prediction_th_ = th; }
    virtual void setNonMaxRadius(double r) { non_maxima_radius_ = r; }
    virtual void setNAnnulus(int n) { n_annulus_ = n; }
    virtual void setNBins(int n) { n_bins_ = n; }

    // hpp:159-176: false when the file cannot be loaded or holds no tree
    virtual bool loadForest(const std::string &path) {
        if (!handle_) return report("loadForest", create_status_);
        int rc = kpl_load_forest_file(handle_, path.c_str());
        if (rc != KPL_OK) {
            PCL_ERROR("[pcl::%s::loadForest] impossible to load random forest with path %s (%s)\n",
                      this->name_.c_str(), path.c_str(), kpl_last_error(handle_));
            return false;
        }
        int ntrees = 0;
        kpl_forest_info(handle_, &ntrees, nullptr, nullptr, nullptr);
        return ntrees != 0;
    }

    // hpp:299-318
    kpl::FeatureMat computePointsForTrainingFeatures(pcl::PointIndicesConstPtr indices) {
        kpl::FeatureMat features;
        if (!this->initCompute()) return features;
        const int m = (int)indices->indices.size();
        const int F = n_annulus_ * n_bins_;
        std::vector<float> buf((size_t)m * F);
        int rc = kpl_compute_features(handle_, &this->input_->points[0].x, sizeof(PointInT),
                                      &normals_->points[0].normal_x, sizeof(NormalT),
                                      (int)this->input_->points.size(), indices->indices.data(), m, buf.data());
        if (this->input_ == this->surface_) this->surface_.reset();
        if (rc != KPL_OK) {
            report("computePointsForTrainingFeatures", rc);
            return features;
        }
#ifdef KPL_USE_OPENCV
        features = cv::Mat(m, F, CV_32F);
        for (int r = 0; r < m; ++r) std::copy(buf.begin() + (size_t)r * F, buf.begin() + (size_t)(r + 1) * F, features.ptr<float>(r));
#else
        features.rows = m;
        features.cols = F;
        features.data.swap(buf);
#endif
        return features;
    }

    // forest responses of the last compute(), one per input point (NaN where not scoreable) -- only
    // when asked for with setKeepScores(true) BEFORE compute(): by default the responses of the
    // keypoints alone come back from the device (no reference counterpart: the reference keeps the
    // responses in a temporary cloud, hpp:182-187)
    void setKeepScores(bool keep) { keep_scores_ = keep; }
    const std::vector<float> &getScores() const {
        if (!keep_scores_)
            PCL_ERROR("[pcl::%s::getScores] empty: call setKeepScores(true) before compute() to keep the response of every point\n",
                      this->name_.c_str());
        return scores_;
    }

    // Neighbor order of the feature loop (hpp:334-359).  By default it follows the search method: a tree set
    // with the inherited setSearchMethod whose getSortedResults() is true (a default-constructed
    // pcl::search::KdTree) selects FLANN's sorted order, no tree (pcl::Keypoint then makes an UNSORTED one,
    // whose traversal order cannot be reproduced) the engine's canonical order.  setSortedSearch overrides it.
    void setSortedSearch(bool sorted) { sorted_search_ = sorted ? 1 : 0; }
    // How the engine walks a neighborhood (KPL_WALK_*, kpl.h): a choice of speed, never of result.  The default
    // (KPL_WALK_AUTO) follows what the handle measured on its earlier calls, or the view's bounding box on a first one.
    void setFeatureWalk(int walk, int lanes_per_point = 2) { if (handle_) kpl_set_feature_walk(handle_, walk, lanes_per_point); }
    // ... and what the last compute() took (KPL_WALK_LANES / KPL_WALK_TWO_PASS; -1: sorted order, or no call yet): a plain
    // read of the handle (kpl_get_last_launch), the per-phase times of kpl_get_timing are left alone
    int getFeatureWalk(int *lanes_per_point = nullptr) const {
        kpl_launch_info li;
        if (!handle_ || kpl_get_last_launch(handle_, &li) != KPL_OK) return -1;
        if (lanes_per_point) *lanes_per_point = li.lanes_per_point;
        return li.walk;
    }
    // the engine's handle behind this detector (no reference counterpart): for callers that also want the preparation steps of
    // the reference's main on the device -- kpl_cloud_resolution, kpl_estimate_normals -- without a second handle (a handle
    // sets up two HIP streams: 8-9 ms each).  Null when no HIP device was found.
    kpl_detector *nativeHandle() const { return handle_; }
    bool getSortedSearch() const { return sorted_search_ >= 0 ? sorted_search_ != 0 : (this->tree_ && this->tree_->getSortedResults()); }
    const char *lastError() const { return handle_ ? kpl_last_error(handle_) : kpl_status_string(create_status_); }

protected:
    bool initCompute() {                                                     // hpp:116-156
        if (!handle_) return report("initCompute", create_status_);
        // pcl::Keypoint::initCompute (hpp:119) restated WITHOUT its search tree: PCL's version allocates a
        // pcl::search::KdTree and builds a FLANN index over the whole cloud on the host at every compute() --
        // tens of milliseconds that this engine, which searches its own index on the device, would never use.
        // What of it survives: pcl::PCLBase::initCompute (the input must be set; the (fake) indices are set up, so
        // getIndices() answers as it does in the reference), the radius / K checks with PCL's own messages, the reset of
        // keypoints_indices_.  tree_ stays whatever the caller set with setSearchMethod (only its getSortedResults() is
        // looked at); getSearchMethod() of a detector nobody gave a tree returns null and searchForNeighbors() is not
        // served -- INTEGRATION.md, section A.
#ifdef KPL_USE_PCL
        if (!pcl::PCLBase<PointInT>::initCompute()) {
            PCL_ERROR("[pcl::%s::initCompute] init failed!\n", this->name_.c_str());
            return false;
        }
#endif
        if (!this->input_) {
            PCL_ERROR("[pcl::%s::initCompute] init failed!\n", this->name_.c_str());
            return false;
        }
        if (!this->surface_) this->surface_ = this->input_;
        if (this->search_radius_ == 0.0 && this->k_ == 0) {
            PCL_ERROR("[pcl::%s::initCompute] Neither radius nor K defined! Set one of them to a positive value (using setRadiusSearch or setKSearch) and then re-run compute ().\n",
                      this->name_.c_str());
            return false;
        }
        if (this->search_radius_ != 0.0 && this->k_ != 0) {
            PCL_ERROR("[pcl::%s::initCompute] Both radius (%f) and K (%d) defined! Set one of them to zero first and then re-run compute ().\n",
                      this->name_.c_str(), this->search_radius_, this->k_);
            return false;
        }
        this->keypoints_indices_.reset(new pcl::PointIndices);
        this->keypoints_indices_->indices.reserve(this->input_->size());
        if (this->surface_ != this->input_) {
            PCL_ERROR("[pcl::%s::initCompute] a search surface different from the input is not supported\n", this->name_.c_str());
            return false;
        }
        if (this->k_ != 0 || !(this->search_radius_ > 0.0)) {
            PCL_ERROR("[pcl::%s::initCompute] only setRadiusSearch(r > 0) is supported\n", this->name_.c_str());
            return false;
        }
        if (!normals_) {                                                     // hpp:125-148
            std::cout << "Computing normals for KPL" << std::endl;
            PointCloudNPtr normals(new PointCloudN());
            const int n = (int)this->surface_->points.size();
            normals->points.resize((size_t)n);
            normals->width = this->surface_->width;
            normals->height = this->surface_->height;
            int rc;
            // both PCL estimators flip the normals towards the sensor origin of their input cloud by default
            // (use_sensor_origin_ = true; pcl::PointCloud::sensor_origin_, the VIEWPOINT of a PCD file)
            const float viewpoint[3] = {this->surface_->sensor_origin_.coeff(0), this->surface_->sensor_origin_.coeff(1),
                                        this->surface_->sensor_origin_.coeff(2)};
            if (this->surface_->isOrganized() &&
                (size_t)this->surface_->width * (size_t)this->surface_->height != this->surface_->points.size()) {
                PCL_ERROR("[pcl::%s::initCompute] organized cloud with width * height != number of points\n", this->name_.c_str());
                return false;
            }
            if (!this->surface_->isOrganized()) {                            // hpp:129-136: pcl::NormalEstimation, radius search
                normals->width = (uint32_t)n;
                normals->height = 1;
                rc = kpl_estimate_normals(handle_, n ? &this->surface_->points[0].x : nullptr, sizeof(PointInT), n, 0,
                                          this->search_radius_, viewpoint,
                                          n ? &normals->points[0].normal_x : nullptr, sizeof(NormalT),
                                          n ? &normals->points[0].curvature : nullptr, sizeof(NormalT));
            } else {                                                         // hpp:138-145: IntegralImageNormalEstimation,
                                                                             // SIMPLE_3D_GRADIENT, smoothing size 5.0
                rc = kpl_estimate_normals_organized(handle_, n ? &this->surface_->points[0].x : nullptr, sizeof(PointInT),
                                                    (int)this->surface_->width, (int)this->surface_->height, 5.0f, viewpoint,
                                                    n ? &normals->points[0].normal_x : nullptr, sizeof(NormalT),
                                                    n ? &normals->points[0].curvature : nullptr, sizeof(NormalT));
            }
            if (rc != KPL_OK) return report("initCompute", rc);
            normals_ = normals;
            if (host_staging_) stage_normals();
        }
        if (normals_->size() != this->surface_->size()) {                   // hpp:149-153
            PCL_ERROR("[pcl::%s::initCompute] normals given, but the number of normals does not match the number of input points!\n", this->name_.c_str());
            return false;
        }
        kpl_params p;
        kpl_default_params(&p);
        p.n_annulus = n_annulus_;
        p.n_bins = n_bins_;
        p.radius_search = this->search_radius_;
        p.non_max_radius = non_maxima_radius_;
        p.prediction_th = prediction_th_;
        p.non_maxima = non_maxima_ ? 1 : 0;
        p.non_maxima_draws_remove = non_maxima_draws_remove_ ? 1 : 0;
        p.non_maxima_draws_threshold = non_maxima_draws_threshold_;
        p.neighbor_order = getSortedSearch() ? KPL_NEIGHBORS_SORTED : KPL_NEIGHBORS_CANONICAL;
        int rc = kpl_set_params(handle_, &p);
        return rc == KPL_OK || report("initCompute", rc);
    }

    virtual void detectKeypoints(PointCloudOut &output) {                   // hpp:179-263
        const int n = (int)this->input_->points.size();
        // indices and responses of the keypoints only come back from the device; the buffers are members
        // (grow-only), like the device scratch of the handle
        if (kp_idx_.size() < (size_t)(n > 0 ? n : 1)) {
            kp_idx_.resize((size_t)(n > 0 ? n : 1));
            kp_score_.resize(kp_idx_.size());
        }
        int count = 0;
        int rc;
        if (host_staging_ && staged_points_ && staged_normals_ && staged_n_ == n && !keep_scores_) {
            scores_.clear();
            rc = kpl_detect_keypoints_staged(handle_, kp_idx_.data(), kp_score_.data(), n, &count);
        } else if (keep_scores_) {
            scores_.assign((size_t)n, 0.0f);
            rc = kpl_detect(handle_, n ? &this->input_->points[0].x : nullptr, sizeof(PointInT),
                            n ? &normals_->points[0].normal_x : nullptr, sizeof(NormalT), n, scores_.data(),
                            kp_idx_.data(), n, &count);
            for (int k = 0; rc == KPL_OK && k < count; ++k) kp_score_[(size_t)k] = scores_[(size_t)kp_idx_[(size_t)k]];
        } else {
            scores_.clear();
            rc = kpl_detect_keypoints(handle_, n ? &this->input_->points[0].x : nullptr, sizeof(PointInT),
                                      n ? &normals_->points[0].normal_x : nullptr, sizeof(NormalT), n,
                                      kp_idx_.data(), kp_score_.data(), n, &count);
        }
        if (rc != KPL_OK) {
            report("detectKeypoints", rc);
            return;
        }
        output.points.clear();
        output.points.resize((size_t)count);
        this->keypoints_indices_->indices.reserve(this->keypoints_indices_->indices.size() + (size_t)count);
        for (int k = 0; k < count; ++k) {
            const PointInT &in = this->input_->points[kp_idx_[k]];
            PointOutT &out = output.points[(size_t)k];
            out.x = in.x;
            out.y = in.y;
            out.z = in.z;
            out.intensity = kp_score_[k];                                    // hpp:284-287
            this->keypoints_indices_->indices.push_back(kp_idx_[k]);
        }
        output.height = 1;                                                   // hpp:258-260
        output.width = static_cast<uint32_t>(output.points.size());
        output.is_dense = non_maxima_ ? true : this->input_->is_dense;       // hpp:193
    }

    // hpp:267-296: the forest response of every point with finite xyz and normal, input order (one
    // device pass with the NMS switched off; valid after initCompute(), like the reference's)
    virtual void runForest(PointCloudOut &output) const {
        const int n = (int)this->input_->points.size();
        kpl_params p, saved;
        if (kpl_get_params(handle_, &saved) != KPL_OK) return;
        p = saved;
        p.non_maxima = 0;
        kpl_set_params(handle_, &p);
        std::vector<float> scores((size_t)(n > 0 ? n : 1));
        std::vector<int> idx((size_t)(n > 0 ? n : 1));
        int count = 0;
        int rc = kpl_detect(handle_, n ? &this->input_->points[0].x : nullptr, sizeof(PointInT),
                            n ? &normals_->points[0].normal_x : nullptr, sizeof(NormalT), n,
                            scores.data(), idx.data(), n, &count);
        kpl_set_params(handle_, &saved);
        if (rc != KPL_OK) {
            report("runForest", rc);
            return;
        }
        for (int k = 0; k < count; ++k) {
            const PointInT &in = this->input_->points[idx[k]];
            PointOutT out;
            out.x = in.x;
            out.y = in.y;
            out.z = in.z;
            out.intensity = scores[idx[k]];
            output.points.push_back(out);
        }
        output.height = 1;
        output.width = static_cast<uint32_t>(output.points.size());
        output.is_dense = true;
    }

    // hpp:321-376: the feature row of one input point
    kpl::FeatureMat computePointFeatures(int point_index) const {
        kpl::FeatureMat features;
        const int F = n_annulus_ * n_bins_;
        std::vector<float> buf((size_t)F);
        int rc = kpl_compute_features(handle_, &this->input_->points[0].x, sizeof(PointInT),
                                      &normals_->points[0].normal_x, sizeof(NormalT),
                                      (int)this->input_->points.size(), &point_index, 1, buf.data());
        if (rc != KPL_OK) {
            report("computePointFeatures", rc);
            return features;
        }
#ifdef KPL_USE_OPENCV
        features = cv::Mat(1, F, CV_32F);
        std::copy(buf.begin(), buf.end(), features.ptr<float>(0));
#else
        features.rows = 1;
        features.cols = F;
        features.data.swap(buf);
#endif
        return features;
    }

    // setHostStaging: packed copies of the points / normals in the engine's pinned buffers
    void stage_points() {
        staged_points_ = staged_normals_ = false;
        if (!handle_ || !this->input_) return;
        const int n = (int)this->input_->points.size();
        void *px = nullptr, *pn = nullptr;
        if (kpl_host_staging(handle_, n, 12, 12, &px, &pn) != KPL_OK) return;
        float *o = static_cast<float *>(px);
        for (int i = 0; i < n; ++i, o += 3) {
            const PointInT &p = this->input_->points[(size_t)i];
            o[0] = p.x;
            o[1] = p.y;
            o[2] = p.z;
        }
        staged_n_ = n;
        staged_points_ = true;
        if (normals_) stage_normals();
    }
    void stage_normals() {
        staged_normals_ = false;
        if (!handle_ || !staged_points_ || !normals_ || (int)normals_->points.size() != staged_n_) return;
        void *px = nullptr, *pn = nullptr;
        if (kpl_host_staging(handle_, staged_n_, 12, 12, &px, &pn) != KPL_OK) return;      // (same size: the same buffers)
        float *o = static_cast<float *>(pn);
        for (int i = 0; i < staged_n_; ++i, o += 3) {
            const NormalT &q = normals_->points[(size_t)i];
            o[0] = q.normal_x;
            o[1] = q.normal_y;
            o[2] = q.normal_z;
        }
        staged_normals_ = true;
    }

    bool report(const char *where, int rc) const {
        PCL_ERROR("[pcl::%s::%s] %s: %s\n", this->name_.c_str(), where, kpl_status_string(rc),
                  handle_ ? kpl_last_error(handle_) : "no HIP device (there is no CPU fallback)");
        return false;
    }

    bool non_maxima_;
    bool non_maxima_draws_remove_;
    float non_maxima_draws_threshold_;
    double prediction_th_;
    double non_maxima_radius_;
    int n_annulus_;
    int n_bins_;
    PointCloudNConstPtr normals_;
    std::vector<float> scores_;
    std::vector<int> kp_idx_;            // grow-only landing buffers of detectKeypoints
    std::vector<float> kp_score_;
    bool keep_scores_ = false;
    bool host_staging_ = false, staged_points_ = false, staged_normals_ = false;
    int staged_n_ = 0;
    int sorted_search_ = -1;             // -1: follow the search method (tree_), 0 / 1: setSortedSearch
    kpl_detector *handle_ = nullptr;
    int create_status_ = KPL_OK;
};

}  // namespace keypoints
}  // namespace pcl
