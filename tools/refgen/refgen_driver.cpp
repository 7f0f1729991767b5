// refgen_driver.cpp -- regenerates the expected outputs of this repo's fixtures with the REFERENCE ITSELF.
//
//   refgen_driver <dir written by tools/refgen/export_inputs.py>
//
// Reads <dir>/manifest.txt, runs every line through pcl::keypoints::KeypointLearningDetector and writes the outputs as raw
// little-endian arrays into <dir>/results/ (+ results/summary.txt); tools/refgen/compare.py holds them against
// tests/golden/*.npz bit for bit.  What each mode calls:
//   detect            the body of /root/reference/src/main_test_detector.cpp:123-187 (one detector, loadForest, setInputCloud,
//                     setNormals, compute) -- twice: setNonMaxima(false) leaves the forest response of every scored point in
//                     the output cloud (impl/KeypointLearning.hpp:189-196), setNonMaxima(true) the keypoints;
//   features          computePointsForTrainingFeatures (impl/KeypointLearning.hpp:299-318) on the listed indices;
//   normals_k         pcl::NormalEstimation, setKSearch (main_test_detector.cpp:162-169);
//   normals_radius    pcl::NormalEstimation, setRadiusSearch (the detector's fallback, hpp:130-137);
//   normals_organized pcl::IntegralImageNormalEstimation, SIMPLE_3D_GRADIENT, smoothing 5 (hpp:138-145).
// sorted=1 hands the detector a pcl::search::KdTree<PointInT>(true) through the inherited setSearchMethod: FLANN then sorts
// every radius result by (distance, index) and the order of the float additions of the histogram no longer depends on the
// layout of its tree -- the one configuration in which a PCL build and this repo can agree in every bit.
//
// ONE source, two builds -- the class API is the contract, so the same driver compiles against either header:
//   * reference build (a machine with PCL 1.8 + OpenCV 3.2; tools/refgen/CMakeLists.txt): "KeypointLearning.h" is found in
//     <reference checkout>/include.  Nothing of the reference is copied here.
//   * -DREFGEN_WITH_KPL (this repo; tests/test_gpu_refgen.py): "KeypointLearning.h" is include/KeypointLearning.h of this
//     repo over libkpl -- the kit's end-to-end self-test, and a demonstration that the drop-in header takes a caller written
//     for the reference as it stands.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#ifdef REFGEN_WITH_KPL
#include "KeypointLearning.h"          // this repo's drop-in (include/), shim types unless -DKPL_USE_PCL
#else
#define PCL_NO_PRECOMPILE
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#include <pcl/io/pcd_io.h>
#include <pcl/search/kdtree.h>
#include <pcl/features/normal_3d.h>
#include <pcl/features/integral_image_normal.h>
#include "KeypointLearning.h"          // the reference's own header (<reference>/include), which pulls in impl/KeypointLearning.hpp
#undef PCL_NO_PRECOMPILE
#endif

typedef pcl::PointXYZ PointInT;        // what the reference's TestDetector instantiates (src/main_test_detector.cpp:93-95)
typedef pcl::Normal PointNormalT;
typedef pcl::PointXYZI KeypointT;
typedef pcl::keypoints::KeypointLearningDetector<PointInT, KeypointT> Detector;

namespace {

typedef std::map<std::string, std::string> Row;

bool parse_manifest(const std::string &path, std::vector<Row> &rows) {
    std::ifstream f(path.c_str());
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string tok;
        Row r;
        while (ss >> tok) {
            const size_t eq = tok.find('=');
            if (eq != std::string::npos) r[tok.substr(0, eq)] = tok.substr(eq + 1);
        }
        if (r.count("id") && r.count("mode")) rows.push_back(r);
    }
    return true;
}

double num(const Row &r, const char *key, double dflt = 0.0) {
    Row::const_iterator it = r.find(key);
    return it == r.end() ? dflt : strtod(it->second.c_str(), NULL);          // (C99 hex floats: exact)
}
std::string str(const Row &r, const char *key) {
    Row::const_iterator it = r.find(key);
    return it == r.end() ? std::string() : it->second;
}
bool viewpoint(const Row &r, float vp[3]) {
    const std::string s = str(r, "viewpoint");
    if (s.empty()) return false;
    const char *p = s.c_str();
    for (int k = 0; k < 3; ++k) {
        char *end = NULL;
        vp[k] = (float)strtod(p, &end);
        p = (*end == ',') ? end + 1 : end;
    }
    return true;
}

template <typename T>
bool write_raw(const std::string &path, const std::vector<T> &v) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const size_t w = v.empty() ? 0 : fwrite(&v[0], sizeof(T), v.size(), f);
    fclose(f);
    return w == v.size();
}

#ifdef REFGEN_WITH_KPL
// the exporter's PCD files: float32 fields, COUNT 1, DATA binary
bool read_pcd_columns(const std::string &path, std::vector<std::string> &fields, std::vector<float> &data, size_t &n,
                      uint32_t &width, uint32_t &height, float vp[3]) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) return false;
    std::string line;
    n = 0;
    width = height = 0;
    vp[0] = vp[1] = vp[2] = 0.f;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "FIELDS") { std::string s; while (ss >> s) fields.push_back(s); }
        else if (key == "WIDTH") ss >> width;
        else if (key == "HEIGHT") ss >> height;
        else if (key == "VIEWPOINT") ss >> vp[0] >> vp[1] >> vp[2];
        else if (key == "POINTS") ss >> n;
        else if (key == "DATA") { std::string kind; ss >> kind; if (kind != "binary") return false; break; }
    }
    data.resize(n * fields.size());
    f.read(reinterpret_cast<char *>(data.data()), (std::streamsize)(data.size() * sizeof(float)));
    return (size_t)f.gcount() == data.size() * sizeof(float);
}
bool load_cloud(const std::string &path, pcl::PointCloud<PointInT> &cloud) {
    std::vector<std::string> fields;
    std::vector<float> d;
    size_t n;
    float vp[3];
    if (!read_pcd_columns(path, fields, d, n, cloud.width, cloud.height, vp) || fields.size() < 3) return false;
    cloud.points.resize(n);
    for (size_t i = 0; i < n; ++i) { cloud.points[i].x = d[i * fields.size()]; cloud.points[i].y = d[i * fields.size() + 1]; cloud.points[i].z = d[i * fields.size() + 2]; }
    for (int k = 0; k < 3; ++k) cloud.sensor_origin_[k] = vp[k];
    cloud.is_dense = false;
    return true;
}
bool load_normals(const std::string &path, pcl::PointCloud<PointNormalT> &normals) {
    std::vector<std::string> fields;
    std::vector<float> d;
    size_t n;
    float vp[3];
    if (!read_pcd_columns(path, fields, d, n, normals.width, normals.height, vp) || fields.size() < 3) return false;
    normals.points.resize(n);
    for (size_t i = 0; i < n; ++i) {
        normals.points[i].normal_x = d[i * fields.size()];
        normals.points[i].normal_y = d[i * fields.size() + 1];
        normals.points[i].normal_z = d[i * fields.size() + 2];
        normals.points[i].curvature = fields.size() > 3 ? d[i * fields.size() + 3] : 0.f;
    }
    return true;
}
#else
bool load_cloud(const std::string &path, pcl::PointCloud<PointInT> &cloud) { return pcl::io::loadPCDFile(path, cloud) == 0; }
bool load_normals(const std::string &path, pcl::PointCloud<PointNormalT> &normals) { return pcl::io::loadPCDFile(path, normals) == 0; }
#endif

// pcl::NormalEstimation / pcl::IntegralImageNormalEstimation the way the reference drives them; both flip towards the
// sensor origin of their input cloud (the VIEWPOINT of the .pcd) by default
bool estimate_normals(const Row &r, const pcl::PointCloud<PointInT>::Ptr &cloud, pcl::PointCloud<PointNormalT> &normals) {
    const std::string mode = str(r, "mode");
#ifdef REFGEN_WITH_KPL
    kpl_detector *h = NULL;
    if (kpl_create(&h, 0) != KPL_OK) return false;
    const int n = (int)cloud->points.size();
    normals.points.resize((size_t)n);
    const float vp[3] = {cloud->sensor_origin_.coeff(0), cloud->sensor_origin_.coeff(1), cloud->sensor_origin_.coeff(2)};
    int rc;
    if (mode == "normals_organized")
        rc = kpl_estimate_normals_organized(h, &cloud->points[0].x, sizeof(PointInT), (int)cloud->width, (int)cloud->height,
                                            (float)num(r, "smoothing", 5.0), vp, &normals.points[0].normal_x, sizeof(PointNormalT),
                                            &normals.points[0].curvature, sizeof(PointNormalT));
    else
        rc = kpl_estimate_normals(h, &cloud->points[0].x, sizeof(PointInT), n, mode == "normals_k" ? (int)num(r, "k", 10) : 0,
                                  num(r, "r_feat"), vp, &normals.points[0].normal_x, sizeof(PointNormalT),
                                  &normals.points[0].curvature, sizeof(PointNormalT));
    if (rc != KPL_OK) fprintf(stderr, "normals: %s\n", kpl_last_error(h));
    kpl_destroy(h);
    return rc == KPL_OK;
#else
    if (mode == "normals_organized") {                                              // hpp:138-145
        pcl::IntegralImageNormalEstimation<PointInT, PointNormalT> ne;
        ne.setNormalEstimationMethod(pcl::IntegralImageNormalEstimation<PointInT, PointNormalT>::SIMPLE_3D_GRADIENT);
        ne.setInputCloud(cloud);
        ne.setNormalSmoothingSize((float)num(r, "smoothing", 5.0));
        ne.compute(normals);
        return true;
    }
    pcl::NormalEstimation<PointInT, PointNormalT> ne;
    ne.setInputCloud(cloud);
    pcl::search::KdTree<PointInT>::Ptr kdtree(new pcl::search::KdTree<PointInT>());
    ne.setSearchMethod(kdtree);
    if (mode == "normals_k") ne.setKSearch((int)num(r, "k", 10));                  // main_test_detector.cpp:162-169
    else ne.setRadiusSearch(num(r, "r_feat"));                                      // hpp:130-137
    ne.compute(normals);
    return true;
#endif
}

struct Summary {
    std::ofstream f;
    void line(const std::string &id, const std::string &what) { f << id << " " << what << "\n"; f.flush(); std::cout << id << " " << what << std::endl; }
};

Detector::Ptr make_detector(const Row &r) {
    Detector::Ptr det(new Detector());
    det->setNAnnulus((int)num(r, "annuli", 5));
    det->setNBins((int)num(r, "bins", 10));
    det->setNonMaxRadius(num(r, "r_nms"));
    det->setNonMaximaDrawsRemove(num(r, "draws_remove") != 0.0);
    det->setNonMaximaDrawsThreshold((float)num(r, "draws_thr"));
    det->setPredictionThreshold(num(r, "thr", 0.5));
    det->setRadiusSearch(num(r, "r_feat"));
    if (num(r, "sorted") != 0.0) {
        pcl::search::KdTree<PointInT>::Ptr tree(new pcl::search::KdTree<PointInT>(true));   // sorted results
        det->setSearchMethod(tree);
    }
    return det;
}

int run_row(const std::string &dir, const Row &r, Summary &sum) {
    const std::string id = str(r, "id"), mode = str(r, "mode"), out = dir + "/results/" + id;
    pcl::PointCloud<PointInT>::Ptr cloud(new pcl::PointCloud<PointInT>());
    if (!load_cloud(dir + "/" + str(r, "cloud"), *cloud)) { sum.line(id, "error=cannot_load_cloud"); return 1; }

    if (mode == "normals_k" || mode == "normals_radius" || mode == "normals_organized") {
        pcl::PointCloud<PointNormalT> normals;
        if (!estimate_normals(r, cloud, normals)) { sum.line(id, "error=normal_estimation_failed"); return 1; }
        std::vector<float> nv, cv;
        for (size_t i = 0; i < normals.points.size(); ++i) {
            nv.push_back(normals.points[i].normal_x); nv.push_back(normals.points[i].normal_y); nv.push_back(normals.points[i].normal_z);
            cv.push_back(normals.points[i].curvature);
        }
        write_raw(out + ".normals.f32", nv);
        write_raw(out + ".curvature.f32", cv);
        std::ostringstream s; s << "normals=" << normals.points.size();
        sum.line(id, s.str());
        return 0;
    }

    pcl::PointCloud<PointNormalT>::Ptr normals(new pcl::PointCloud<PointNormalT>());
    if (!load_normals(dir + "/" + str(r, "normals"), *normals)) { sum.line(id, "error=cannot_load_normals"); return 1; }

    if (mode == "features") {
        Detector::Ptr det = make_detector(r);
        det->setInputCloud(cloud);
        det->setNormals(normals);
        pcl::PointIndices::Ptr idx(new pcl::PointIndices());
        std::ifstream q((dir + "/" + str(r, "query")).c_str());
        int v;
        while (q >> v) idx->indices.push_back(v);
        // cv::Mat in the reference build, kpl::FeatureMatrix here: rows / cols / ptr<float>(row) in both
        auto feats = det->computePointsForTrainingFeatures(idx);
        std::vector<float> fv;
        for (int row = 0; row < feats.rows; ++row) fv.insert(fv.end(), feats.ptr<float>(row), feats.ptr<float>(row) + feats.cols);
        write_raw(out + ".features.f32", fv);
        std::ostringstream s; s << "rows=" << feats.rows << " cols=" << feats.cols;
        sum.line(id, s.str());
        return feats.rows == (int)idx->indices.size() ? 0 : 1;
    }

    if (mode == "detect") {
        // pass 1: the forest response of every scored point (non-finite points / normals are skipped by runForest,
        // hpp:277: the output is COMPACTED, compare.py maps it back through the finite mask of the inputs)
        Detector::Ptr det = make_detector(r);
        const bool loaded = det->loadForest(dir + "/" + str(r, "forest"));       // cv::ml::RTrees::load: the YAML dialect check
        sum.line(id, std::string("forest_accepted=") + (loaded ? "yes" : "no"));
        if (!loaded) return 1;
        det->setInputCloud(cloud);
        det->setNormals(normals);
        det->setNonMaxima(false);
        pcl::PointCloud<KeypointT> response;
        det->compute(response);
        std::vector<float> sv;
        for (size_t i = 0; i < response.points.size(); ++i) sv.push_back(response.points[i].intensity);
        write_raw(out + ".scores.f32", sv);
        // pass 2: the keypoints (a fresh detector: one detector, one compute(), like the reference's main)
        Detector::Ptr det2 = make_detector(r);
        if (!det2->loadForest(dir + "/" + str(r, "forest"))) return 1;
        det2->setInputCloud(cloud);
        det2->setNormals(normals);
        det2->setNonMaxima(true);
        pcl::PointCloud<KeypointT> keypoints;
        det2->compute(keypoints);
        std::vector<int32_t> kv;
        std::vector<float> ks;
        pcl::PointIndicesConstPtr kp = det2->getKeypointsIndices();
        for (size_t i = 0; i < kp->indices.size(); ++i) kv.push_back(kp->indices[i]);
        for (size_t i = 0; i < keypoints.points.size(); ++i) ks.push_back(keypoints.points[i].intensity);
        write_raw(out + ".keypoints.i32", kv);
        write_raw(out + ".kp_scores.f32", ks);
        std::ostringstream s; s << "scored=" << sv.size() << " keypoints=" << kv.size();
        sum.line(id, s.str());
        return 0;
    }
    sum.line(id, "error=unknown_mode");
    return 1;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <dir of export_inputs.py> [run id ...]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    std::vector<Row> rows;
    if (!parse_manifest(dir + "/manifest.txt", rows)) { fprintf(stderr, "cannot read %s/manifest.txt\n", dir.c_str()); return 2; }
    Summary sum;
    sum.f.open((dir + "/results/summary.txt").c_str(), argc > 2 ? std::ios::app : std::ios::trunc);
#ifdef REFGEN_WITH_KPL
    sum.line("#", "engine=libkpl (this repo's drop-in header; NOT a reference run)");
#else
    sum.line("#", "engine=reference (PCL + OpenCV)");
#endif
    int failed = 0;
    for (size_t k = 0; k < rows.size(); ++k) {
        bool wanted = argc <= 2;
        for (int a = 2; a < argc; ++a) wanted = wanted || str(rows[k], "id") == argv[a];
        if (wanted) failed += run_row(dir, rows[k], sum);
    }
    return failed ? 1 : 0;
}
