// TestDetector -- counterpart of /root/reference/src/main_test_detector.cpp on top of the drop-in
// header include/KeypointLearning.h (libkpl, MI355X).  Same option names (:58-67): pathCloud,
// pathRF, pathKP, radiusFeatures, radiusNMS, threshold, flipNormals, subSampling, leaf; plus
// annuli / bins (were #defines, :105-106), radiusInMr (radii given in units of the cloud
// resolution) and json.  No viewer (:191-210 dropped).
//
// Preparation as in the reference main: PCD load (:143) and UniformSampling (:145-157) on the
// host; cloud resolution (for --radiusInMr) and the k = 10 normals (:162-169) on the device
// through libkpl (kpl_cloud_resolution, kpl_estimate_normals); flip (:173-179) on the host.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "KeypointLearning.h"
#include "pcd_io.h"

typedef pcl::PointXYZ PointInT;
typedef pcl::Normal PointNormalT;
typedef pcl::PointXYZI KeypointT;

using kpl_io::load_pcd;
using kpl_io::save_pcd_ascii;

namespace {

struct Options {
    std::map<std::string, std::string> kv;
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    std::string str(const std::string &k, const std::string &d) const { return has(k) ? kv.at(k) : d; }
    double num(const std::string &k, double d) const { return has(k) ? atof(kv.at(k).c_str()) : d; }
};

const char *kUsage =
    "Allowed options:\n"
    "  -h [ --help ]                 produce help message\n"
    "  --flipNormals                 If present flip normals, some dataset needs normal re-orientation.\n"
    "  --subSampling                 If present, subsample cloud with leaf.\n"
    "  --leaf arg                    Leaf size for subsampling.\n"
    "  --pathCloud arg               Path to dataset (.pcd, fields x y z [normal_x normal_y normal_z]).\n"
    "  --pathRF arg                  Path to Random Forest (OpenCV YAML, optionally gzipped).\n"
    "  --pathKP arg                  Path for keypoints point cloud.\n"
    "  --radiusFeatures arg (=20)    Radius for features computation.\n"
    "  --radiusNMS arg (=4)          Radius for non maxima suppresion.\n"
    "  -t [ --threshold ] arg (=0.85) Threshold for random forest prediction.\n"
    "  --annuli arg (=5)             Number of annuli (reference: #define ANNULI 5).\n"
    "  --bins arg (=10)              Number of bins (reference: #define BINS 10).\n"
    "  --radiusInMr                  radiusFeatures / radiusNMS / leaf are multiples of the cloud resolution.\n"
    "  --device arg (=0)             HIP device ordinal.\n"
    "  --json                        print one JSON line with counts and timings.\n"
    "  --printResolution             print the cloud resolution (mean 2nd-NN distance) and exit.\n"
    "  --detectorNormals             do not pass normals: the detector estimates them itself (radius search\n"
    "                                with radiusFeatures, as the reference's initCompute does).\n"
    "  --hostStaging                 setHostStaging(true): the setters copy cloud and normals into pinned buffers of the\n"
    "                                engine, compute() uploads them by DMA.\n"
    "  --sortedSearch                hand the detector a sorted search tree (setSearchMethod(pcl::search::KdTree(true))):\n"
    "                                neighbors in ascending (distance, index) order instead of the engine's canonical order.\n"
    "  --walk arg (=auto)            how the engine walks a neighborhood: auto, lanes2, lanes4, twopass2, twopass4\n"
    "                                (setFeatureWalk; a choice of speed, the results are the same bits).\n";

bool parse(int argc, char **argv, Options &o) {
    static const char *flags[] = {"help", "flipNormals", "subSampling", "radiusInMr", "json", "printResolution", "detectorNormals", "checkProtected", "sortedSearch", "hostStaging"};
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h") a = "--help";
        if (a == "-t") a = "--threshold";
        if (a.rfind("--", 0) != 0) { fprintf(stderr, "unrecognised option '%s'\n%s", a.c_str(), kUsage); return false; }
        a = a.substr(2);
        std::string val;
        size_t eq = a.find('=');
        bool has_val = eq != std::string::npos;
        if (has_val) { val = a.substr(eq + 1); a = a.substr(0, eq); }
        bool is_flag = std::find_if(std::begin(flags), std::end(flags), [&](const char *f) { return a == f; }) != std::end(flags);
        if (!is_flag && !has_val) {
            if (i + 1 >= argc) { fprintf(stderr, "the required argument for option '--%s' is missing\n%s", a.c_str(), kUsage); return false; }
            val = argv[++i];
        }
        o.kv[a] = val;
    }
    if (o.has("help")) { printf("%s", kUsage); return false; }
    if (o.has("subSampling") && !o.has("leaf")) { printf("Subsampling needs leaf.\n"); return false; }
    return true;
}

// ---- PCD (ascii / binary; float32 fields) -------------------------------------------------------
// exact voxel key: 21 bits per axis around 0 (voxels further out than 2^20 would alias; clouds that
// large relative to the leaf are not expected here)
long long voxel_key(long long x, long long y, long long z) {
    const long long m = (1LL << 21) - 1, o = 1LL << 20;
    return (((x + o) & m) << 42) | (((y + o) & m) << 21) | ((z + o) & m);
}

// the preparation steps that run on the device, on the DETECTOR's own handle (nativeHandle(): one handle, one set-up, per process)
struct Prep {
    kpl_detector *h = nullptr;
    explicit Prep(kpl_detector *handle) : h(handle) {
        if (!h) fprintf(stderr, "no HIP device\n");
    }
    // computeCloudResolution, /root/reference/include/impl/point_cloud_utilities.hpp:120-151
    bool resolution(const pcl::PointCloud<PointInT> &cloud, double &mr) const {
        const int n = (int)cloud.size();
        int rc = kpl_cloud_resolution(h, n ? &cloud.points[0].x : nullptr, sizeof(PointInT), n, &mr);
        if (rc != KPL_OK) fprintf(stderr, "cloud resolution: %s\n", kpl_last_error(h));
        return rc == KPL_OK;
    }
    // pcl::NormalEstimation with setKSearch(10) -- main_test_detector.cpp:162-169; flipped towards the sensor
    // origin of the cloud (the PCD's VIEWPOINT; NormalEstimation::setInputCloud takes it over by default)
    bool normals(const pcl::PointCloud<PointInT> &cloud, int k, pcl::PointCloud<PointNormalT> &out) const {
        const int n = (int)cloud.size();
        out.clear();
        out.points.resize((size_t)n);
        out.width = (uint32_t)n;
        out.height = 1;
        const float viewpoint[3] = {cloud.sensor_origin_.coeff(0), cloud.sensor_origin_.coeff(1), cloud.sensor_origin_.coeff(2)};
        int rc = kpl_estimate_normals(h, n ? &cloud.points[0].x : nullptr, sizeof(PointInT), n, k, 0.0, viewpoint,
                                      n ? &out.points[0].normal_x : nullptr, sizeof(PointNormalT),
                                      n ? &out.points[0].curvature : nullptr, sizeof(PointNormalT));
        if (rc != KPL_OK) fprintf(stderr, "normal estimation: %s\n", kpl_last_error(h));
        return rc == KPL_OK;
    }
};

// pcl::UniformSampling: one point per leaf-sized voxel, the one closest to the voxel centre
void uniform_sampling(pcl::PointCloud<PointInT> &cloud, double leaf) {
    std::unordered_map<long long, std::pair<double, int>> best;
    for (int i = 0; i < (int)cloud.size(); ++i) {
        const PointInT &p = cloud[i];
        if (!pcl::isFinite(p)) continue;
        const long long x = (long long)std::floor(p.x / leaf), y = (long long)std::floor(p.y / leaf), z = (long long)std::floor(p.z / leaf);
        const double cx = (x + 0.5) * leaf, cy = (y + 0.5) * leaf, cz = (z + 0.5) * leaf;
        const double d = (p.x - cx) * (p.x - cx) + (p.y - cy) * (p.y - cy) + (p.z - cz) * (p.z - cz);
        auto it = best.find(voxel_key(x, y, z));
        if (it == best.end() || d < it->second.first) best[voxel_key(x, y, z)] = {d, i};
    }
    std::vector<int> keep;
    for (auto &kv : best) keep.push_back(kv.second.second);
    std::sort(keep.begin(), keep.end());
    pcl::PointCloud<PointInT> out;
    for (int i : keep) out.push_back(cloud[i]);
    out.sensor_origin_ = cloud.sensor_origin_;
    cloud = out;
}

// --checkProtected: a subclass reaches the protected members the reference declares
// (runForest, computePointFeatures: /root/reference/include/KeypointLearning.h:170-177)
struct ProbeDetector : pcl::keypoints::KeypointLearningDetector<PointInT, KeypointT> {
    using Base = pcl::keypoints::KeypointLearningDetector<PointInT, KeypointT>;
    ProbeDetector(int device) : Base(0.5, true, true, 0.0, 5, 10, device) {}
    // returns the number of mismatches between runForest / computePointFeatures and compute()
    int check(const pcl::PointCloud<PointInT> &cloud) {
        pcl::PointCloud<KeypointT> kp, all;
        this->setKeepScores(true);
        this->compute(kp);
        const std::vector<float> scores = this->getScores();
        if (!this->initCompute()) return -1;
        this->runForest(all);
        size_t k = 0;
        int bad = 0;
        for (size_t i = 0; i < scores.size(); ++i) {
            if (std::isnan(scores[i])) continue;
            if (k >= all.size() || all[k].x != cloud[i].x || memcmp(&all[k].intensity, &scores[i], 4) != 0) ++bad;
            ++k;
        }
        if (k != all.size()) ++bad;
        pcl::PointIndices::Ptr one(new pcl::PointIndices());
        for (int i : {0, (int)cloud.size() / 2, (int)cloud.size() - 1}) {
            one->indices.assign(1, i);
            kpl::FeatureMat a = this->computePointFeatures(i);
            kpl::FeatureMat b = this->computePointsForTrainingFeatures(one);
            this->setInputCloud(this->input_);       // (the reference resets surface_ in the call above)
            if (a.rows != 1 || a.cols != b.cols || memcmp(a.data.data(), b.data.data(), sizeof(float) * (size_t)a.cols) != 0) ++bad;
        }
        return bad;
    }
};

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

int main(int argc, char **argv) {
    Options vm;
    if (!parse(argc, argv, vm)) return 0;

    float radius_nms = (float)vm.num("radiusNMS", 4.0);
    float radius_features = (float)vm.num("radiusFeatures", 20.0);
    const float threshold = (float)vm.num("threshold", 0.85);
    const std::string path_rf = vm.str("pathRF", "../../../data/forest/SHOT-LaserScanner.yaml.gz");
    const std::string path_cloud = vm.str("pathCloud", "../../../data/point_cloud_test/cheff001.pcd");
    const int annuli = (int)vm.num("annuli", 5), bins = (int)vm.num("bins", 10);
    const bool json = vm.has("json");

    const int device = (int)vm.num("device", 0);

    // create detector (:123-130)
    pcl::keypoints::KeypointLearningDetector<PointInT, KeypointT>::Ptr detector(
        new pcl::keypoints::KeypointLearningDetector<PointInT, KeypointT>(0.5, true, true, 0.0, 5, 10, device));
    Prep prep(detector->nativeHandle());
    if (!prep.h) return -1;
    if (vm.has("printResolution")) {
        pcl::PointCloud<PointInT> c;
        pcl::PointCloud<PointNormalT> nn;
        double mr = 0.0;
        if (!load_pcd(path_cloud, c, nn) || !prep.resolution(c, mr)) return -1;
        printf("%.17g\n", mr);
        return 0;
    }

    detector->setNAnnulus(annuli);
    detector->setNBins(bins);
    detector->setNonMaxima(true);
    detector->setNonMaximaDrawsRemove(false);
    detector->setPredictionThreshold(threshold);
    if (detector->loadForest(path_rf)) {
        if (!json) printf("Detector created.\n");
    } else {
        return -1;
    }

    // load and subsample point cloud (:142-157)
    pcl::PointCloud<PointInT>::Ptr cloud(new pcl::PointCloud<PointInT>());
    pcl::PointCloud<PointNormalT>::Ptr normals(new pcl::PointCloud<PointNormalT>());
    if (!load_pcd(path_cloud, *cloud, *normals)) return -1;
    auto t_prep = std::chrono::steady_clock::now();
    double mr = 0.0;
    float leaf = (float)vm.num("leaf", 0.0);
    if (vm.has("radiusInMr")) {
        if (!prep.resolution(*cloud, mr)) return -1;
        radius_features = (float)(radius_features * mr);
        radius_nms = (float)(radius_nms * mr);
        leaf = (float)(leaf * mr);
    }
    if (vm.has("subSampling")) {
        uniform_sampling(*cloud, leaf);
        normals->clear();
    }
    if (!json) printf("Point cloud loaded\n");

    // Compute normals (:162-169) unless the file carried them
    const bool own_normals = vm.has("detectorNormals");
    if (!own_normals && normals->size() != cloud->size() && !prep.normals(*cloud, 10, *normals)) return -1;
    if (!json && !own_normals) printf("Normals Computed\n");
    if (vm.has("flipNormals") && !own_normals) {                                             // :172-179
        if (!json) printf("Flipping \n");
        for (auto &n : normals->points) { n.normal_x *= -1; n.normal_y *= -1; n.normal_z *= -1; }
    }
    const double prep_s = seconds_since(t_prep);

    if (vm.has("checkProtected")) {
        ProbeDetector probe(device);
        probe.setNAnnulus(annuli); probe.setNBins(bins); probe.setNonMaxima(true); probe.setNonMaximaDrawsRemove(false);
        probe.setPredictionThreshold(threshold); probe.setNonMaxRadius(radius_nms); probe.setRadiusSearch(radius_features);
        if (!probe.loadForest(path_rf)) return -1;
        probe.setInputCloud(cloud);
        probe.setNormals(normals);
        const int bad = probe.check(*cloud);
        printf("checkProtected: %d mismatches\n", bad);
        return bad == 0 ? 0 : 2;
    }

    detector->setNonMaxRadius(radius_nms);
    detector->setRadiusSearch(radius_features);
    if (vm.has("sortedSearch"))          // the inherited pcl::Keypoint::setSearchMethod; KdTree's constructor default is sorted = true
        detector->setSearchMethod(pcl::search::KdTree<PointInT>::Ptr(new pcl::search::KdTree<PointInT>(true)));
    if (vm.has("hostStaging")) detector->setHostStaging(true);
    if (vm.has("walk")) {
        const std::string w = vm.str("walk", "auto");
        if (w == "auto") detector->setFeatureWalk(KPL_WALK_AUTO);
        else if (w == "lanes2") detector->setFeatureWalk(KPL_WALK_LANES, 2);
        else if (w == "lanes4") detector->setFeatureWalk(KPL_WALK_LANES, 4);
        else if (w == "twopass2") detector->setFeatureWalk(KPL_WALK_TWO_PASS, 2);
        else if (w == "twopass4") detector->setFeatureWalk(KPL_WALK_TWO_PASS, 4);
        else { fprintf(stderr, "the argument ('%s') for option '--walk' is invalid\n", w.c_str()); return -1; }
    }
    auto t_set = std::chrono::steady_clock::now();
    detector->setInputCloud(cloud);
    if (!own_normals) detector->setNormals(normals);          // else: impl/KeypointLearning.hpp:125-148
    const double set_s = seconds_since(t_set);

    // detect keypoints (:186-187)
    pcl::PointCloud<KeypointT>::Ptr keypoint(new pcl::PointCloud<KeypointT>());
    auto t0 = std::chrono::steady_clock::now();
    detector->compute(*keypoint);
    const double first_s = seconds_since(t0);
    t0 = std::chrono::steady_clock::now();
    detector->compute(*keypoint);                  // second call: scratch buffers already sized
    double warm_s = seconds_since(t0);
    for (int rep = 0; rep < 5; ++rep) {            // steady state: the best of a few more
        t0 = std::chrono::steady_clock::now();
        detector->compute(*keypoint);
        warm_s = std::min(warm_s, seconds_since(t0));
    }
    if (!json) printf("Keypoint computed\n");

    if (vm.has("pathKP")) save_pcd_ascii(vm.str("pathKP", ""), *keypoint);
    int walk_lanes = 0;
    const int walk_taken = detector->getFeatureWalk(&walk_lanes);
    if (json)
        printf("{\"points\": %zu, \"keypoints\": %zu, \"mr\": %.9g, \"radiusFeatures\": %.9g, \"radiusNMS\": %.9g, "
               "\"threshold\": %.9g, \"annuli\": %d, \"bins\": %d, \"prepare_s\": %.6f, \"set_s\": %.6f, \"compute_first_s\": %.6f, "
               "\"compute_s\": %.6f, \"walk\": \"%s\", \"lanes_per_point\": %d}\n",
               cloud->size(), keypoint->size(), mr, radius_features, radius_nms, threshold, annuli, bins, prep_s, set_s, first_s, warm_s,
               walk_taken == KPL_WALK_TWO_PASS ? "two-pass" : walk_taken == KPL_WALK_LANES ? "lanes" : "sorted", walk_lanes);
    else
        printf("%zu keypoints out of %zu points (compute: %.3f ms)\nDONE\n", keypoint->size(), cloud->size(), warm_s * 1e3);
    return 0;
}
