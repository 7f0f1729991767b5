// kpl_pcl_shim.h -- the few PCL names include/KeypointLearning.h needs, for builds WITHOUT PCL.
//
// This is not a PCL re-implementation and not a stand-in used to build the reference: it only
// lets the drop-in class of this repo (and its TestDetector) compile on machines that have no
// PCL, with the same spellings a PCL user writes.  When real PCL is available, define
// KPL_USE_PCL and this file is not included at all.
//
// Layouts follow PCL 1.8: PointXYZ = 16 B (x, y, z, pad), Normal = 32 B (normal_x..z, pad,
// curvature, pad), PointXYZI = 32 B (x, y, z, pad, intensity, pad) -- what
// /root/reference/src/main_test_detector.cpp:93-95 instantiates the detector with.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#define PCL_ERROR(...) std::fprintf(stderr, __VA_ARGS__)

namespace pcl {

struct alignas(16) PointXYZ {
    float x = 0.f, y = 0.f, z = 0.f, _pad = 1.f;
    PointXYZ() = default;
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

struct alignas(16) PointXYZI {
    float x = 0.f, y = 0.f, z = 0.f, _pad = 1.f;
    float intensity = 0.f;
    float _pad2[3] = {0.f, 0.f, 0.f};
};

struct alignas(16) Normal {
    float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, _pad = 0.f;
    float curvature = 0.f;
    float _pad2[3] = {0.f, 0.f, 0.f};
};

template <typename PointT>
inline bool isFinite(const PointT &p) {
    return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z);
}
template <>
inline bool isFinite<Normal>(const Normal &n) {
    return std::isfinite(n.normal_x) && std::isfinite(n.normal_y) && std::isfinite(n.normal_z);
}

// pcl::PointCloud::sensor_origin_ is an Eigen::Vector4f; the drop-in class reads it with coeff(i) only
struct SensorOrigin {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    float coeff(int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
};

template <typename PointT>
class PointCloud {
public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;
    SensorOrigin sensor_origin_;          // the VIEWPOINT of a PCD file: where normal estimation flips to
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void push_back(const PointT &p) {
        points.push_back(p);
        width = (uint32_t)points.size();
        height = 1;
    }
    void reserve(size_t n) { points.reserve(n); }
    void clear() {
        points.clear();
        width = height = 0;
    }
    bool isOrganized() const { return height > 1; }
    PointT &operator[](size_t i) { return points[i]; }
    const PointT &operator[](size_t i) const { return points[i]; }
};

struct PointIndices {
    typedef std::shared_ptr<PointIndices> Ptr;
    typedef std::shared_ptr<const PointIndices> ConstPtr;
    std::vector<int> indices;
};
typedef PointIndices::Ptr PointIndicesPtr;
typedef PointIndices::ConstPtr PointIndicesConstPtr;

// pcl::search::Search / pcl::search::KdTree as far as setSearchMethod needs them here: the engine searches
// its own index on the device and only asks a tree whether its results are sorted (PCL 1.8:
// pcl::search::KdTree<PointT>(bool sorted = true); pcl::Keypoint makes a KdTree(false) when none is set)
namespace search {
template <typename PointT>
class Search {
public:
    typedef std::shared_ptr<Search<PointT>> Ptr;
    typedef std::shared_ptr<const Search<PointT>> ConstPtr;
    explicit Search(const std::string &name = "", bool sorted = false) : sorted_results_(sorted), name_(name) {}
    virtual ~Search() {}
    virtual void setSortedResults(bool sorted_results) { sorted_results_ = sorted_results; }
    virtual bool getSortedResults() { return sorted_results_; }

protected:
    bool sorted_results_;
    std::string name_;
};
template <typename PointT>
class KdTree : public Search<PointT> {
public:
    typedef std::shared_ptr<KdTree<PointT>> Ptr;
    typedef std::shared_ptr<const KdTree<PointT>> ConstPtr;
    explicit KdTree(bool sorted = true) : Search<PointT>("KdTree", sorted) {}
};
}  // namespace search

// pcl::Keypoint as far as the detector and TestDetector use it
// (setInputCloud / setSearchSurface / setSearchMethod / setRadiusSearch / setKSearch / compute / getKeypointsIndices).
template <typename PointInT, typename PointOutT>
class Keypoint {
public:
    typedef PointCloud<PointInT> PointCloudIn;
    typedef typename PointCloudIn::ConstPtr PointCloudInConstPtr;
    typedef PointCloud<PointOutT> PointCloudOut;
    typedef pcl::search::Search<PointInT> KdTree;
    typedef typename KdTree::Ptr KdTreePtr;

    Keypoint() : keypoints_indices_(new PointIndices) {}
    virtual ~Keypoint() {}

    virtual void setInputCloud(const PointCloudInConstPtr &cloud) { input_ = cloud; }
    virtual void setSearchSurface(const PointCloudInConstPtr &cloud) { surface_ = cloud; }
    void setSearchMethod(const KdTreePtr &tree) { tree_ = tree; }
    KdTreePtr getSearchMethod() { return tree_; }
    void setRadiusSearch(double radius) { search_radius_ = radius; }
    double getRadiusSearch() const { return search_radius_; }
    void setKSearch(int k) { k_ = k; }
    PointIndicesConstPtr getKeypointsIndices() const { return keypoints_indices_; }

    void compute(PointCloudOut &output) {
        if (!initCompute()) {
            PCL_ERROR("[pcl::%s::compute] initCompute failed!\n", name_.c_str());
            return;
        }
        detectKeypoints(output);
        if (input_ == surface_) surface_.reset();
    }

protected:
    virtual bool initCompute() {
        if (!input_) return false;
        if (!surface_) surface_ = input_;
        if (search_radius_ != 0.0 && k_ != 0) {
            PCL_ERROR("[pcl::%s::initCompute] Both radius and K defined!\n", name_.c_str());
            return false;
        }
        if (search_radius_ == 0.0 && k_ == 0) {
            PCL_ERROR("[pcl::%s::initCompute] Neither radius nor K defined!\n", name_.c_str());
            return false;
        }
        keypoints_indices_.reset(new PointIndices);
        keypoints_indices_->indices.reserve(input_->size());
        return true;
    }
    virtual void detectKeypoints(PointCloudOut &output) = 0;

    std::string name_;
    PointCloudInConstPtr input_, surface_;
    KdTreePtr tree_;
    double search_radius_ = 0.0;
    int k_ = 0;
    PointIndicesPtr keypoints_indices_;
};

}  // namespace pcl
