// DetectViews -- a set of views scored on every GPU of the node, from plain C++ over the C-ABI.
//
// What a caller that sweeps a data set writes around detector.compute() (the reference walks its views one by
// one in one process, e.g. /root/reference/src/main_train_detector.cpp:285-447).  Here (BASELINE.json configs[2],
// SURVEY.md 8(e)): the views are independent, so they are dealt round robin to the devices (view k -> device
// k mod D, like keypoint-learning_amd/dist.py shard()); ONE host thread per device gives every view of its share
// its own handle, keeps its points and pcl::Normal records in HBM (hipMalloc'd by THIS program, libkpl never owns
// them), gets normals from kpl_estimate_normals_device straight into those records (main_test_detector.cpp:162-169,
// k = 10) and scores up to 8 views per kpl_compute_batch_keypoints_device call, two batches in flight on two
// streams.  No data-path exchange between devices; the one collective is the gather of the results: every device
// packs the keypoint lists of its views ([count][indices][responses] per view) and contributes them to ONE
// ncclAllGather (RCCL over xGMI; a communicator per device, ncclCommInitAll).  Device 0's copy of the gathered
// buffer is what the keypoint files and the JSON lines are written from.
//
//   DetectViews --pathRF forest.yaml.gz --radiusFeatures 6 --radiusNMS 4 [--radiusInMr] [-t 0.85] [--annuli 5]
//               [--bins 10] [--flipNormals] [--sortedSearch] [--devices all|N] [--rounds R] [--pathKP prefix]
//               view0.pcd view1.pcd ...
// One JSON line per view, then one summary line ({"devices": ...}).  --rounds R repeats the scoring + gather R
// times and reports the makespan per round (views resident, like bench.py).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
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
#define CHECK_NCCL(call)                                                                             \
    do {                                                                                             \
        ncclResult_t e_ = (call);                                                                    \
        if (e_ != ncclSuccess) {                                                                     \
            fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(e_));                              \
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

struct Options {
    std::string path_rf, path_kp;
    double r_feat = 20.0, r_nms = 4.0, thr = 0.85;
    int annuli = 5, bins = 10, rounds = 1;
    bool in_mr = false, flip = false, sorted = false;
};

struct View {
    std::string path;
    pcl::PointCloud<PointInT> cloud;
    pcl::PointCloud<PointNormalT> file_normals;
    kpl_detector *h = nullptr;
    PointInT *d_xyz = nullptr;          // 16-byte records, as PCL lays them out
    PointNormalT *d_nrm = nullptr;      // 32-byte records
    int n = 0;
};

// what one device thread owns
struct Worker {
    int device = 0, rank = 0, nranks = 1;
    std::vector<View *> views;          // its share, in view order
    int slots = 0, cap = 0;             // packed result: slots x (1 + 2 cap) ints per device
    int *d_send = nullptr, *d_recv = nullptr;
    hipStream_t st[2] = {nullptr, nullptr};
    ncclComm_t comm = nullptr;
    double seconds = 0.0;               // timed rounds
    int rc = 0;
};

size_t slot_ints(int cap) { return (size_t)1 + 2 * (size_t)cap; }

// The device threads agree on success BEFORE every collective: a thread that failed on the way (a KPL error on one view,
// a second RETRY, a HIP error) must not leave the others waiting in ncclAllGather for a rank that never comes.
// agree(ok) returns true iff every thread of the round said ok; all of them then skip or enter the collective together.
class Rendezvous {
  public:
    explicit Rendezvous(int parties) : parties_(parties) {}
    bool agree(bool ok) {
        std::unique_lock<std::mutex> lock(m_);
        all_ok_ = (arrived_ == 0 ? true : all_ok_) && ok;
        const unsigned long gen = generation_;
        if (++arrived_ == parties_) {
            result_ = all_ok_;
            arrived_ = 0;
            ++generation_;
            cv_.notify_all();
            return result_;
        }
        cv_.wait(lock, [&] { return generation_ != gen; });
        return result_;
    }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    const int parties_;
    int arrived_ = 0;
    unsigned long generation_ = 0;
    bool all_ok_ = true, result_ = true;
};

int prepare(Worker &w, const Options &o) {
    CHECK_HIP(hipSetDevice(w.device));
    CHECK_HIP(hipStreamCreateWithFlags(&w.st[0], hipStreamNonBlocking));
    CHECK_HIP(hipStreamCreateWithFlags(&w.st[1], hipStreamNonBlocking));
    for (View *vp : w.views) {
        View &v = *vp;
        const size_t nn = (size_t)(v.n > 0 ? v.n : 1);
        if (kpl_create(&v.h, w.device) != KPL_OK) { fprintf(stderr, "no HIP device %d\n", w.device); return 1; }
        CHECK_KPL(v.h, kpl_load_forest_file(v.h, o.path_rf.c_str()));
        double mr = 1.0;
        if (o.in_mr) CHECK_KPL(v.h, kpl_cloud_resolution(v.h, v.n ? &v.cloud.points[0].x : nullptr, sizeof(PointInT), v.n, &mr));
        kpl_params p;
        kpl_default_params(&p);
        p.n_annulus = o.annuli;
        p.n_bins = o.bins;
        p.radius_search = (float)(o.r_feat * mr);                     // the reference mains keep radii in float
        p.non_max_radius = (float)(o.r_nms * mr);
        p.prediction_th = (float)o.thr;
        p.non_maxima = 1;
        p.non_maxima_draws_remove = 0;                                // main_test_detector.cpp:128
        p.neighbor_order = o.sorted ? KPL_NEIGHBORS_SORTED : KPL_NEIGHBORS_CANONICAL;
        CHECK_KPL(v.h, kpl_set_params(v.h, &p));
        CHECK_HIP(hipMalloc((void **)&v.d_xyz, nn * sizeof(PointInT)));
        CHECK_HIP(hipMalloc((void **)&v.d_nrm, nn * sizeof(PointNormalT)));
        if (v.n) CHECK_HIP(hipMemcpy(v.d_xyz, v.cloud.points.data(), (size_t)v.n * sizeof(PointInT), hipMemcpyHostToDevice));
        CHECK_HIP(hipDeviceSynchronize());        // (the copy above ran on the null stream; w.st[] are non-blocking streams)
        CHECK_KPL(v.h, kpl_bind_cloud_device(v.h, v.d_xyz, sizeof(PointInT), v.d_nrm, sizeof(PointNormalT), v.n));
        if ((int)v.file_normals.size() == v.n && v.n) {
            CHECK_HIP(hipMemcpy(v.d_nrm, v.file_normals.points.data(), (size_t)v.n * sizeof(PointNormalT), hipMemcpyHostToDevice));
        } else {
            const float viewpoint[3] = {v.cloud.sensor_origin_.coeff(0), v.cloud.sensor_origin_.coeff(1), v.cloud.sensor_origin_.coeff(2)};
            for (int attempt = 0; attempt < 2; ++attempt) {           // a first-time size may have to grow the cell tables once
                CHECK_KPL(v.h, kpl_estimate_normals_device(v.h, 10, 0.0, viewpoint, &v.d_nrm->normal_x, sizeof(PointNormalT),
                                                           &v.d_nrm->curvature, sizeof(PointNormalT), w.st[0]));
                const int rc = kpl_sync_status(v.h, w.st[0]);
                if (rc == KPL_OK) break;
                if (rc != KPL_ERR_RETRY || attempt == 1) { fprintf(stderr, "normals: %s\n", kpl_last_error(v.h)); return 1; }
            }
        }
        if (o.flip && v.n) {                                           // main_test_detector.cpp:172-179, on the host for brevity
            std::vector<PointNormalT> tmp((size_t)v.n);
            CHECK_HIP(hipMemcpy(tmp.data(), v.d_nrm, (size_t)v.n * sizeof(PointNormalT), hipMemcpyDeviceToHost));
            for (auto &q : tmp) { q.normal_x *= -1; q.normal_y *= -1; q.normal_z *= -1; }
            CHECK_HIP(hipMemcpy(v.d_nrm, tmp.data(), (size_t)v.n * sizeof(PointNormalT), hipMemcpyHostToDevice));
        }
    }
    CHECK_HIP(hipMalloc((void **)&w.d_send, sizeof(int) * slot_ints(w.cap) * (size_t)w.slots));
    CHECK_HIP(hipMalloc((void **)&w.d_recv, sizeof(int) * slot_ints(w.cap) * (size_t)w.slots * (size_t)w.nranks));
    CHECK_HIP(hipMemset(w.d_send, 0, sizeof(int) * slot_ints(w.cap) * (size_t)w.slots));
    // the null stream of the copies and the clear above is NOT ordered against the non-blocking streams the scoring runs
    // on, and hipMemset returns before it is done (tests/csrc/memset_probe.cpp)
    CHECK_HIP(hipDeviceSynchronize());
    return 0;
}

// one round: every view of this device scored (batches of up to 8, alternating streams), results written by the
// engine straight into the packed send buffer ...
int score(Worker &w, bool first) {
    CHECK_HIP(hipSetDevice(w.device));
    const int nv = (int)w.views.size();
    // first round: checked until TWO rounds in a row came back clean.  A handle grows what a view needs when it first sees
    // it -- the cell tables, in sorted-search mode the key array (its need can rise once more: the chunk tails of the wave
    // kernel are counted like keys and depend on the order the waves ran in) -- and after its first clean round it knows the
    // neighborhood size of the view and may switch to the two-pass walk, whose word list then grows once too: none of this
    // may happen in the timed rounds, which are not read back
    int clean = 0;
    for (int attempt = 0; attempt < 10 && clean < (first ? 2 : 1); ++attempt) {
        for (int b0 = 0, bi = 0; b0 < nv; b0 += 8, ++bi) {
            const int cnt = std::min(8, nv - b0);
            kpl_detector *hs[8];
            int *kps[8], *counts[8], caps[8];
            float *kscores[8];
            for (int j = 0; j < cnt; ++j) {
                int *slot = w.d_send + slot_ints(w.cap) * (size_t)(b0 + j);
                hs[j] = w.views[(size_t)(b0 + j)]->h;
                counts[j] = slot;
                kps[j] = slot + 1;
                kscores[j] = reinterpret_cast<float *>(slot + 1 + w.cap);
                caps[j] = w.cap;
            }
            CHECK_KPL(hs[0], kpl_compute_batch_keypoints_device(hs, cnt, kps, kscores, caps, counts, w.st[bi & 1]));
        }
        if (!first) break;
        bool retry = false;
        for (int k = 0; k < nv; ++k) {
            const int rc = kpl_sync_status(w.views[(size_t)k]->h, w.st[(k / 8) & 1]);
            if (rc == KPL_ERR_RETRY) retry = true;
            else if (rc != KPL_OK) { fprintf(stderr, "%s: %s\n", w.views[(size_t)k]->path.c_str(), kpl_last_error(w.views[(size_t)k]->h)); return 1; }
        }
        clean = retry ? 0 : clean + 1;
    }
    if (first && clean < 2) { fprintf(stderr, "device %d: the views keep asking for larger tables (10 rounds)\n", w.device); return 1; }
    CHECK_HIP(hipStreamSynchronize(w.st[1]));                          // the gather is enqueued behind stream 0: wait for the other one
    return 0;
}

// ... then the node-wide gather -- entered by every device thread or by none (Rendezvous)
int score_and_gather(Worker &w, bool first, Rendezvous &rv) {
    const int rc = score(w, first);
    if (!rv.agree(rc == 0)) return rc ? rc : 1;                        // some device failed: nobody enters the collective
    // ... and every thread learns how the collective went on EVERY device before anyone goes on to the next round: a thread
    // whose ncclAllGather or stream wait failed would leave the loop, and the others would wait forever in the next
    // round's agree() for a party that never arrives
    bool ok = ncclAllGather(w.d_send, w.d_recv, slot_ints(w.cap) * (size_t)w.slots, ncclInt32, w.comm, w.st[0]) == ncclSuccess;
    if (!ok) fprintf(stderr, "device %d: ncclAllGather failed\n", w.device);
    if (ok && hipStreamSynchronize(w.st[0]) != hipSuccess) {
        fprintf(stderr, "device %d: the gather's stream failed\n", w.device);
        ok = false;
    }
    if (!rv.agree(ok)) return 1;
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    Options o;
    std::string devices = "1";
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--pathRF") o.path_rf = next();
        else if (a == "--pathKP") o.path_kp = next();
        else if (a == "--radiusFeatures") o.r_feat = atof(next());
        else if (a == "--radiusNMS") o.r_nms = atof(next());
        else if (a == "-t" || a == "--threshold") o.thr = atof(next());
        else if (a == "--annuli") o.annuli = atoi(next());
        else if (a == "--bins") o.bins = atoi(next());
        else if (a == "--rounds") o.rounds = std::max(1, atoi(next()));
        else if (a == "--devices") devices = next();
        else if (a == "--radiusInMr") o.in_mr = true;
        else if (a == "--flipNormals") o.flip = true;
        else if (a == "--sortedSearch") o.sorted = true;
        else files.push_back(a);
    }
    if (o.path_rf.empty() || files.empty()) {
        fprintf(stderr, "usage: DetectViews --pathRF forest --radiusFeatures r --radiusNMS r [options] [--devices all|N] view.pcd ...\n");
        return 2;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) { fprintf(stderr, "no HIP device (there is no CPU fallback)\n"); return 1; }
    int ndev = devices == "all" ? visible : atoi(devices.c_str());
    if (ndev < 1 || ndev > visible) { fprintf(stderr, "--devices %s: %d device(s) visible\n", devices.c_str(), visible); return 2; }
    ndev = std::min(ndev, (int)files.size());

    std::vector<View> views(files.size());
    int cap = 1;
    for (size_t k = 0; k < files.size(); ++k) {
        views[k].path = files[k];
        if (!kpl_io::load_pcd(files[k], views[k].cloud, views[k].file_normals)) return 1;
        views[k].n = (int)views[k].cloud.size();
        cap = std::max(cap, views[k].n);
    }
    std::vector<Worker> workers((size_t)ndev);
    std::vector<int> devlist((size_t)ndev);
    for (int d = 0; d < ndev; ++d) {
        workers[(size_t)d].device = devlist[(size_t)d] = d;
        workers[(size_t)d].rank = d;
        workers[(size_t)d].nranks = ndev;
        workers[(size_t)d].cap = cap;
    }
    for (size_t k = 0; k < views.size(); ++k) workers[k % (size_t)ndev].views.push_back(&views[k]);      // round robin
    const int slots = (int)((views.size() + (size_t)ndev - 1) / (size_t)ndev);
    for (Worker &w : workers) w.slots = slots;
    std::vector<ncclComm_t> comms((size_t)ndev);
    CHECK_NCCL(ncclCommInitAll(comms.data(), ndev, devlist.data()));
    for (int d = 0; d < ndev; ++d) workers[(size_t)d].comm = comms[(size_t)d];

    // one host thread per device: prepare, one checked round, then the timed rounds
    Rendezvous rv(ndev);
    auto body = [&](Worker &w) {
        w.rc = prepare(w, o);
        if (!rv.agree(w.rc == 0)) {                                    // a device could not be prepared: nobody scores
            if (!w.rc) w.rc = 1;
            return;
        }
        w.rc = score_and_gather(w, true, rv);
    };
    {
        std::vector<std::thread> th;
        for (Worker &w : workers) th.emplace_back(body, std::ref(w));
        for (auto &t : th) t.join();
    }
    for (Worker &w : workers) if (w.rc) return w.rc;
    double makespan = 0.0;
    if (o.rounds > 1) {
        auto timed = [&](Worker &w) {
            const auto t0 = std::chrono::steady_clock::now();
            // (a failed round fails on every device -- score_and_gather agrees before AND after the collective --, so all
            // threads leave the loop together)
            for (int r = 0; r < o.rounds && !w.rc; ++r) w.rc = score_and_gather(w, false, rv);
            w.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        std::vector<std::thread> th;
        for (Worker &w : workers) th.emplace_back(timed, std::ref(w));
        for (auto &t : th) t.join();
        for (Worker &w : workers) {
            if (w.rc) return w.rc;
            makespan = std::max(makespan, w.seconds / o.rounds);
        }
    }

    // ---- everything below reads DEVICE 0's copy of the gathered buffer --------------------------------------
    Worker &w0 = workers[0];
    CHECK_HIP(hipSetDevice(w0.device));
    std::vector<int> all(slot_ints(cap) * (size_t)slots * (size_t)ndev);
    CHECK_HIP(hipMemcpy(all.data(), w0.d_recv, sizeof(int) * all.size(), hipMemcpyDeviceToHost));
    long long total_points = 0, total_kp = 0;
    for (size_t k = 0; k < views.size(); ++k) {
        const View &v = views[k];
        const int *slot = all.data() + slot_ints(cap) * ((k % (size_t)ndev) * (size_t)slots + k / (size_t)ndev);
        const int nk = slot[0];
        if (nk < 0 || nk > v.n) { fprintf(stderr, "%s: bad keypoint count %d\n", v.path.c_str(), nk); return 1; }
        long long checksum = 0;
        pcl::PointCloud<KeypointT> out;
        for (int j = 0; j < nk; ++j) {
            const int idx = slot[1 + j];
            checksum += idx;
            KeypointT q;
            q.x = v.cloud[(size_t)idx].x; q.y = v.cloud[(size_t)idx].y; q.z = v.cloud[(size_t)idx].z;
            memcpy(&q.intensity, &slot[1 + cap + j], sizeof(float));
            out.push_back(q);
        }
        if (!o.path_kp.empty()) kpl_io::save_pcd_ascii(o.path_kp + std::to_string(k) + ".pcd", out);
        printf("{\"view\": \"%s\", \"device\": %d, \"points\": %d, \"keypoints\": %d, \"index_checksum\": %lld}\n", v.path.c_str(),
               (int)(k % (size_t)ndev), v.n, nk, checksum);
        total_points += v.n;
        total_kp += nk;
    }
    std::string per_device = "[";
    for (const Worker &w : workers) {
        char buf[64];
        snprintf(buf, sizeof(buf), "%s%.4f", w.rank ? ", " : "", o.rounds > 1 ? w.seconds / o.rounds * 1e3 : 0.0);
        per_device += buf;
    }
    per_device += "]";
    printf("{\"devices\": %d, \"views\": %zu, \"views_per_device\": %d, \"points\": %lld, \"keypoints\": %lld, "
           "\"exchange\": \"one ncclAllGather of %d x %zu int32 per round\", \"rounds\": %d, \"makespan_ms_per_round\": %.4f, "
           "\"device_ms_per_round\": %s, \"Mpoints_per_s\": %.2f}\n",
           ndev, views.size(), slots, total_points, total_kp, ndev * slots, slot_ints(cap), o.rounds, makespan * 1e3,
           per_device.c_str(), makespan > 0 ? total_points / makespan / 1e6 : 0.0);
    for (Worker &w : workers) {
        (void)hipSetDevice(w.device);
        for (View *v : w.views) { (void)hipFree(v->d_xyz); (void)hipFree(v->d_nrm); kpl_destroy(v->h); }
        (void)hipFree(w.d_send); (void)hipFree(w.d_recv);
        (void)hipStreamDestroy(w.st[0]); (void)hipStreamDestroy(w.st[1]);
        ncclCommDestroy(w.comm);
    }
    return 0;
}
