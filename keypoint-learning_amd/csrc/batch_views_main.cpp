// DetectViews -- several views per call through the C-ABI, device-resident, from plain C++.
//
// What a caller that sweeps a data set writes around detector.compute() (the reference walks its views
// one by one, e.g. /root/reference/src/main_train_detector.cpp:285-447): here every view gets its own
// handle, its points and pcl::Normal records live in HBM (hipMalloc'd by THIS program, libkpl never
// owns them), normals come from kpl_estimate_normals_device straight into those records
// (main_test_detector.cpp:162-169, k = 10), and up to 8 views go through ONE
// kpl_compute_batch_device call.  Prints one JSON line per view.
//
//   DetectViews --pathRF forest.yaml.gz --radiusFeatures 6 --radiusNMS 4 [--radiusInMr] [-t 0.85]
//               [--annuli 5] [--bins 10] [--flipNormals] [--pathKP prefix] view1.pcd view2.pcd ...
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pcd_io.h"

using kpl_io::KeypointT;
using kpl_io::PointInT;
using kpl_io::PointNormalT;

namespace {

#define CHECK_HIP(call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                               \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)
#define CHECK_KPL(h, call)                                                                           \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != KPL_OK) {                                                                         \
            fprintf(stderr, "%s: %s (%s)\n", #call, kpl_status_string(rc_), kpl_last_error(h));     \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

struct View {
    std::string path;
    pcl::PointCloud<PointInT> cloud;
    pcl::PointCloud<PointNormalT> file_normals;
    kpl_detector *h = nullptr;
    PointInT *d_xyz = nullptr;          // 16-byte records, as PCL lays them out
    PointNormalT *d_nrm = nullptr;      // 32-byte records
    float *d_scores = nullptr;
    int *d_kp = nullptr, *d_count = nullptr;
    int n = 0;
};

}  // namespace

int main(int argc, char **argv) {
    std::string path_rf, path_kp;
    double r_feat = 20.0, r_nms = 4.0, thr = 0.85;
    int annuli = 5, bins = 10;
    bool in_mr = false, flip = false;
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--pathRF") path_rf = next();
        else if (a == "--pathKP") path_kp = next();
        else if (a == "--radiusFeatures") r_feat = atof(next());
        else if (a == "--radiusNMS") r_nms = atof(next());
        else if (a == "-t" || a == "--threshold") thr = atof(next());
        else if (a == "--annuli") annuli = atoi(next());
        else if (a == "--bins") bins = atoi(next());
        else if (a == "--radiusInMr") in_mr = true;
        else if (a == "--flipNormals") flip = true;
        else files.push_back(a);
    }
    if (path_rf.empty() || files.empty() || files.size() > 8) {
        fprintf(stderr, "usage: DetectViews --pathRF forest --radiusFeatures r --radiusNMS r [options] view.pcd ... (1 to 8 views)\n");
        return 2;
    }
    std::vector<View> views(files.size());
    const float origin[3] = {0.0f, 0.0f, 0.0f};
    for (size_t k = 0; k < files.size(); ++k) {
        View &v = views[k];
        v.path = files[k];
        if (!kpl_io::load_pcd(v.path, v.cloud, v.file_normals)) return 1;
        v.n = (int)v.cloud.size();
        const size_t nn = (size_t)(v.n > 0 ? v.n : 1);
        if (kpl_create(&v.h, 0) != KPL_OK) { fprintf(stderr, "no HIP device\n"); return 1; }
        CHECK_KPL(v.h, kpl_load_forest_file(v.h, path_rf.c_str()));
        double mr = 1.0;
        if (in_mr) CHECK_KPL(v.h, kpl_cloud_resolution(v.h, v.n ? &v.cloud.points[0].x : nullptr, sizeof(PointInT), v.n, &mr));
        kpl_params p;
        kpl_default_params(&p);
        p.n_annulus = annuli;
        p.n_bins = bins;
        p.radius_search = (float)(r_feat * mr);                       // the reference mains keep radii in float
        p.non_max_radius = (float)(r_nms * mr);
        p.prediction_th = (float)thr;
        p.non_maxima = 1;
        p.non_maxima_draws_remove = 0;                                // main_test_detector.cpp:128
        CHECK_KPL(v.h, kpl_set_params(v.h, &p));
        CHECK_HIP(hipMalloc((void **)&v.d_xyz, nn * sizeof(PointInT)));
        CHECK_HIP(hipMalloc((void **)&v.d_nrm, nn * sizeof(PointNormalT)));
        CHECK_HIP(hipMalloc((void **)&v.d_scores, nn * sizeof(float)));
        CHECK_HIP(hipMalloc((void **)&v.d_kp, nn * sizeof(int)));
        CHECK_HIP(hipMalloc((void **)&v.d_count, sizeof(int)));
        if (v.n) CHECK_HIP(hipMemcpy(v.d_xyz, v.cloud.points.data(), (size_t)v.n * sizeof(PointInT), hipMemcpyHostToDevice));
        CHECK_KPL(v.h, kpl_bind_cloud_device(v.h, v.d_xyz, sizeof(PointInT), v.d_nrm, sizeof(PointNormalT), v.n));
        if ((int)v.file_normals.size() == v.n && v.n) {
            CHECK_HIP(hipMemcpy(v.d_nrm, v.file_normals.points.data(), (size_t)v.n * sizeof(PointNormalT), hipMemcpyHostToDevice));
        } else {
            for (int attempt = 0; attempt < 2; ++attempt) {           // a first-time size may have to grow the cell tables once
                CHECK_KPL(v.h, kpl_estimate_normals_device(v.h, 10, 0.0, origin, &v.d_nrm->normal_x, sizeof(PointNormalT),
                                                           &v.d_nrm->curvature, sizeof(PointNormalT), nullptr));
                const int rc = kpl_sync_status(v.h, nullptr);
                if (rc == KPL_OK) break;
                if (rc != KPL_ERR_RETRY || attempt == 1) { fprintf(stderr, "normals: %s\n", kpl_last_error(v.h)); return 1; }
            }
        }
        if (flip && v.n) {                                             // main_test_detector.cpp:172-179, on the host for brevity
            std::vector<PointNormalT> tmp((size_t)v.n);
            CHECK_HIP(hipMemcpy(tmp.data(), v.d_nrm, (size_t)v.n * sizeof(PointNormalT), hipMemcpyDeviceToHost));
            for (auto &q : tmp) { q.normal_x *= -1; q.normal_y *= -1; q.normal_z *= -1; }
            CHECK_HIP(hipMemcpy(v.d_nrm, tmp.data(), (size_t)v.n * sizeof(PointNormalT), hipMemcpyHostToDevice));
        }
    }
    // ---- all views in one call ----------------------------------------------------------------
    const int count = (int)views.size();
    std::vector<kpl_detector *> hs;
    std::vector<float *> scores;
    std::vector<int *> kps, counts;
    std::vector<int> caps;
    for (View &v : views) { hs.push_back(v.h); scores.push_back(v.d_scores); kps.push_back(v.d_kp); counts.push_back(v.d_count); caps.push_back(v.n); }
    for (int attempt = 0; attempt < 2; ++attempt) {
        CHECK_KPL(hs[0], kpl_compute_batch_device(hs.data(), count, scores.data(), kps.data(), caps.data(), counts.data(), nullptr));
        bool retry = false;
        for (View &v : views) {
            const int rc = kpl_sync_status(v.h, nullptr);
            if (rc == KPL_ERR_RETRY && attempt == 0) retry = true;
            else if (rc != KPL_OK) { fprintf(stderr, "%s: %s\n", v.path.c_str(), kpl_last_error(v.h)); return 1; }
        }
        if (!retry) break;
    }
    for (size_t k = 0; k < views.size(); ++k) {
        View &v = views[k];
        int nk = 0;
        CHECK_HIP(hipMemcpy(&nk, v.d_count, sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> idx((size_t)(nk > 0 ? nk : 1));
        std::vector<float> sc((size_t)(v.n > 0 ? v.n : 1));
        if (nk > 0) CHECK_HIP(hipMemcpy(idx.data(), v.d_kp, (size_t)nk * sizeof(int), hipMemcpyDeviceToHost));
        if (v.n > 0) CHECK_HIP(hipMemcpy(sc.data(), v.d_scores, (size_t)v.n * sizeof(float), hipMemcpyDeviceToHost));
        long long checksum = 0;
        pcl::PointCloud<KeypointT> out;
        for (int j = 0; j < nk; ++j) {
            checksum += idx[j];
            KeypointT q;
            q.x = v.cloud[idx[j]].x; q.y = v.cloud[idx[j]].y; q.z = v.cloud[idx[j]].z; q.intensity = sc[idx[j]];
            out.push_back(q);
        }
        if (!path_kp.empty()) kpl_io::save_pcd_ascii(path_kp + std::to_string(k) + ".pcd", out);
        printf("{\"view\": \"%s\", \"points\": %d, \"keypoints\": %d, \"index_checksum\": %lld}\n", v.path.c_str(), v.n, nk, checksum);
        hipFree(v.d_xyz); hipFree(v.d_nrm); hipFree(v.d_scores); hipFree(v.d_kp); hipFree(v.d_count);
        kpl_destroy(v.h);
    }
    return 0;
}
