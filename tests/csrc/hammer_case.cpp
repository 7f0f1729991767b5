// hammer_case.cpp -- one saved view through libkpl again and again, every result compared bit for bit with the expected
// one, optionally while a second handle keeps the GPU busy with a 200 k-point view on another stream (timing perturbation).
// No Python in the loop.  Written for the round-3 fuzz event (tests/golden/fuzz_31337.npz -> tools/case_blob.py -> blob).
//
//   hammer_case <case.blob> <seconds> [host|device] [load] [fresh] [syncdev]
//     host     kpl_detect on pageable host buffers (the path the fuzz tool uses)            (default)
//     device   kpl_bind_cloud_device + kpl_compute_device, results read back through pinned memory
//     load     a second thread runs kpl_compute_device on a synthetic 200 k-point view on its own stream meanwhile
//     fresh    host mode: the cloud is copied into freshly malloc'ed arrays before every call (what numpy does)
//     syncdev  hipDeviceSynchronize between the calls
//     newhandle host mode: a NEW handle for every call (kpl_create ... kpl_destroy): every call goes through the growth of the
//              handle's tables -- for a view whose grid has 2.6e8 cells that is a 1 GB table, allocated and cleared, and
//              the KPL_ERR_RETRY that follows (the round-3 / round-4 fuzz events were such views)
// Exit code 0 = every iteration identical to the expectation, 1 = mismatches (reported, first ones dumped), 2 = setup error.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kpl.h"

#define HIPCHECK(x)                                                                           \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)
#define KPLCHECK(h, x)                                                                        \
    do {                                                                                      \
        int rc_ = (x);                                                                        \
        if (rc_ != KPL_OK) {                                                                  \
            fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, kpl_last_error(h));                   \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

struct Case {
    int n, A, B, nms, draws, sorted, ntrees, nnodes, var_count, n_kp;
    double r, rn, thr, dthr;
    std::vector<float> xyz, nrm, thrs, scores;
    std::vector<int> root, var, left, right, kp;
    std::vector<double> value;
};

template <class T> static bool rd(FILE *f, std::vector<T> &v, size_t count) {
    v.resize(count);
    return count == 0 || fread(v.data(), sizeof(T), count, f) == count;
}

static bool load_case(const char *path, Case &c) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    int hdr[10];
    double par[4];
    bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "KPLCASE1", 8) == 0 && fread(hdr, 4, 10, f) == 10 && fread(par, 8, 4, f) == 4;
    if (ok) {
        c.n = hdr[0]; c.A = hdr[1]; c.B = hdr[2]; c.nms = hdr[3]; c.draws = hdr[4]; c.sorted = hdr[5];
        c.ntrees = hdr[6]; c.nnodes = hdr[7]; c.var_count = hdr[8]; c.n_kp = hdr[9];
        c.r = par[0]; c.rn = par[1]; c.thr = par[2]; c.dthr = par[3];
        ok = rd(f, c.xyz, 3 * (size_t)c.n) && rd(f, c.nrm, 3 * (size_t)c.n) && rd(f, c.root, (size_t)c.ntrees) &&
             rd(f, c.var, (size_t)c.nnodes) && rd(f, c.thrs, (size_t)c.nnodes) && rd(f, c.left, (size_t)c.nnodes) &&
             rd(f, c.right, (size_t)c.nnodes) && rd(f, c.value, (size_t)c.nnodes) && rd(f, c.scores, (size_t)c.n) &&
             rd(f, c.kp, (size_t)c.n_kp);
    }
    fclose(f);
    return ok;
}

static void configure(kpl_detector *h, const Case &c) {
    kpl_params p;
    kpl_default_params(&p);
    p.n_annulus = c.A;
    p.n_bins = c.B;
    p.radius_search = c.r;
    p.non_max_radius = c.rn;
    p.prediction_th = c.thr;
    p.non_maxima = c.nms;
    p.non_maxima_draws_remove = c.draws;
    p.non_maxima_draws_threshold = (float)c.dthr;
    p.neighbor_order = c.sorted ? KPL_NEIGHBORS_SORTED : KPL_NEIGHBORS_CANONICAL;
    KPLCHECK(h, kpl_set_params(h, &p));
    KPLCHECK(h, kpl_load_forest_arrays(h, c.ntrees, c.nnodes, c.var_count, c.root.data(), c.var.data(), c.thrs.data(),
                                       c.left.data(), c.right.data(), c.value.data()));
}

// NaN == NaN whatever the payload, everything else by bits
static size_t diff_scores(const float *a, const float *b, int n, int *first) {
    size_t d = 0;
    for (int i = 0; i < n; ++i) {
        const bool na = a[i] != a[i], nb = b[i] != b[i];
        uint32_t x, y;
        memcpy(&x, a + i, 4);
        memcpy(&y, b + i, 4);
        if (na != nb || (!na && x != y)) {
            if (d == 0 && first) *first = i;
            ++d;
        }
    }
    return d;
}

// ---- the co-running load: a synthetic 200 k-point range-image-like surface, 5 x 6 histogram, 10 random trees
static std::atomic<bool> g_stop{false};
static std::atomic<long> g_load_calls{0};

static void load_thread() {
    const int nx = 500, ny = 400, n = nx * ny;
    std::vector<float> xyz(3 * (size_t)n), nrm(3 * (size_t)n);
    const float s = 1.0f / nx;
    for (int j = 0; j < ny; ++j)
        for (int i = 0; i < nx; ++i) {
            const float x = i * s, y = j * s;
            const float z = 0.05f * sinf(9.0f * x) * cosf(7.0f * y) + 0.02f * sinf(31.0f * x + 17.0f * y);
            const float zx = 0.45f * cosf(9.0f * x) * cosf(7.0f * y) + 0.62f * cosf(31.0f * x + 17.0f * y);
            const float zy = -0.35f * sinf(9.0f * x) * sinf(7.0f * y) + 0.34f * cosf(31.0f * x + 17.0f * y);
            const float inv = 1.0f / sqrtf(zx * zx + zy * zy + 1.0f);
            const size_t k = 3 * ((size_t)j * nx + i);
            xyz[k] = x; xyz[k + 1] = y; xyz[k + 2] = z;
            nrm[k] = -zx * inv; nrm[k + 1] = -zy * inv; nrm[k + 2] = inv;
        }
    // 10 random trees over 30 variables, depth <= 10, leaves 0 / 1
    std::vector<int> root, var, left, right;
    std::vector<float> thr;
    std::vector<double> value;
    uint32_t rng = 12345u;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    struct Todo { int node, depth; };
    for (int t = 0; t < 10; ++t) {
        std::vector<Todo> todo;
        auto new_node = [&]() { var.push_back(-1); thr.push_back(0.f); left.push_back(-1); right.push_back(-1); value.push_back(0.0); return (int)var.size() - 1; };
        root.push_back(new_node());
        todo.push_back({root.back(), 0});
        while (!todo.empty()) {
            const Todo cur = todo.back();
            todo.pop_back();
            if (cur.depth >= 10 || (cur.depth > 3 && next() % 4 == 0)) {
                value[cur.node] = (double)(next() & 1);
                continue;
            }
            var[cur.node] = (int)(next() % 30);
            thr[cur.node] = (float)(next() % 1000) * 0.0006f;
            const int l = new_node(), r = new_node();
            left[cur.node] = l;
            right[cur.node] = r;
            todo.push_back({l, cur.depth + 1});
            todo.push_back({r, cur.depth + 1});
        }
    }
    kpl_detector *h = nullptr;
    if (kpl_create(&h, 0) != KPL_OK) {
        fprintf(stderr, "load thread: kpl_create failed\n");
        return;
    }
    kpl_params p;
    kpl_default_params(&p);
    p.n_annulus = 5;
    p.n_bins = 6;
    p.radius_search = 6.0 * s;
    p.non_max_radius = 4.0 * s;
    p.prediction_th = 0.85;
    p.non_maxima_draws_remove = 0;
    KPLCHECK(h, kpl_set_params(h, &p));
    KPLCHECK(h, kpl_load_forest_arrays(h, 10, (int)var.size(), 30, root.data(), var.data(), thr.data(), left.data(), right.data(), value.data()));
    void *dx, *dn, *ds, *dk, *dc;
    HIPCHECK(hipMalloc(&dx, xyz.size() * 4));
    HIPCHECK(hipMalloc(&dn, nrm.size() * 4));
    HIPCHECK(hipMalloc(&ds, (size_t)n * 4));
    HIPCHECK(hipMalloc(&dk, (size_t)n * 4));
    HIPCHECK(hipMalloc(&dc, 16));
    HIPCHECK(hipMemcpy(dx, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dn, nrm.data(), nrm.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    KPLCHECK(h, kpl_bind_cloud_device(h, dx, 12, dn, 12, n));
    while (!g_stop.load()) {
        for (int k = 0; k < 4; ++k) KPLCHECK(h, kpl_compute_device(h, (float *)ds, (int *)dk, n, (int *)dc, st));
        int rc = kpl_sync_status(h, st);
        if (rc != KPL_OK && rc != KPL_ERR_RETRY) {
            fprintf(stderr, "load thread: %s\n", kpl_last_error(h));
            break;
        }
        g_load_calls += 4;
    }
    HIPCHECK(hipStreamSynchronize(st));
    kpl_destroy(h);
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: hammer_case <case.blob> <seconds> [host|device] [load] [fresh] [syncdev]\n");
        return 2;
    }
    Case c;
    if (!load_case(argv[1], c)) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    const double budget = atof(argv[2]);
    bool device = false, load = false, fresh = false, syncdev = false, newhandle = false;
    for (int a = 3; a < argc; ++a) {
        const std::string s = argv[a];
        device |= s == "device";
        load |= s == "load";
        fresh |= s == "fresh";
        syncdev |= s == "syncdev";
        newhandle |= s == "newhandle";
    }
    kpl_detector *h = nullptr;
    if (kpl_create(&h, 0) != KPL_OK) {
        fprintf(stderr, "kpl_create failed\n");
        return 2;
    }
    configure(h, c);
    const int n = c.n;
    std::thread loader;
    if (load) loader = std::thread(load_thread);

    void *dx = nullptr, *dn = nullptr, *ds = nullptr, *dk = nullptr, *dc = nullptr;
    float *p_scores = nullptr;
    int *p_kp = nullptr, *p_cnt = nullptr;
    hipStream_t st = nullptr;
    if (device) {
        HIPCHECK(hipMalloc(&dx, 12 * (size_t)n + 16));
        HIPCHECK(hipMalloc(&dn, 12 * (size_t)n + 16));
        HIPCHECK(hipMalloc(&ds, 4 * (size_t)n + 16));
        HIPCHECK(hipMalloc(&dk, 4 * (size_t)n + 16));
        HIPCHECK(hipMalloc(&dc, 16));
        HIPCHECK(hipMemcpy(dx, c.xyz.data(), 12 * (size_t)n, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(dn, c.nrm.data(), 12 * (size_t)n, hipMemcpyHostToDevice));
        HIPCHECK(hipHostMalloc((void **)&p_scores, 4 * (size_t)n + 16, hipHostMallocDefault));
        HIPCHECK(hipHostMalloc((void **)&p_kp, 4 * (size_t)n + 16, hipHostMallocDefault));
        HIPCHECK(hipHostMalloc((void **)&p_cnt, 16, hipHostMallocDefault));
        HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        KPLCHECK(h, kpl_bind_cloud_device(h, dx, 12, dn, 12, n));
    }
    std::vector<float> scores((size_t)n + 1);
    std::vector<int> kp((size_t)n + 1);
    long iters = 0, bad = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    while (elapsed() < budget) {
        for (int rep = 0; rep < (newhandle ? 4 : 64); ++rep, ++iters) {
            int cnt = -12345;
            const float *got_scores;
            const int *got_kp;
            if (device) {
                for (int attempt = 0; attempt < 2; ++attempt) {
                    KPLCHECK(h, kpl_compute_device(h, (float *)ds, (int *)dk, n, (int *)dc, st));
                    HIPCHECK(hipMemcpyAsync(p_scores, ds, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
                    HIPCHECK(hipMemcpyAsync(p_kp, dk, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
                    HIPCHECK(hipMemcpyAsync(p_cnt, dc, 4, hipMemcpyDeviceToHost, st));
                    const int rc = kpl_sync_status(h, st);
                    if (rc == KPL_OK) break;
                    if (rc != KPL_ERR_RETRY) KPLCHECK(h, rc);
                }
                cnt = p_cnt[0];
                got_scores = p_scores;
                got_kp = p_kp;
            } else if (newhandle) {
                kpl_detector *hn = nullptr;
                if (kpl_create(&hn, 0) != KPL_OK) {
                    fprintf(stderr, "kpl_create failed\n");
                    return 2;
                }
                configure(hn, c);
                KPLCHECK(hn, kpl_detect(hn, c.xyz.data(), 12, c.nrm.data(), 12, n, scores.data(), kp.data(), n, &cnt));
                kpl_destroy(hn);
                got_scores = scores.data();
                got_kp = kp.data();
            } else if (fresh) {
                float *x = (float *)malloc(12 * (size_t)n + 4), *m = (float *)malloc(12 * (size_t)n + 4);
                float *so = (float *)malloc(4 * (size_t)n + 4);
                memcpy(x, c.xyz.data(), 12 * (size_t)n);
                memcpy(m, c.nrm.data(), 12 * (size_t)n);
                KPLCHECK(h, kpl_detect(h, x, 12, m, 12, n, so, kp.data(), n, &cnt));
                memcpy(scores.data(), so, 4 * (size_t)n);
                free(x);
                free(m);
                free(so);
                got_scores = scores.data();
                got_kp = kp.data();
            } else {
                KPLCHECK(h, kpl_detect(h, c.xyz.data(), 12, c.nrm.data(), 12, n, scores.data(), kp.data(), n, &cnt));
                got_scores = scores.data();
                got_kp = kp.data();
            }
            int first = -1;
            const size_t ds_bad = diff_scores(got_scores, c.scores.data(), n, &first);
            const bool kp_bad = cnt != c.n_kp || (cnt > 0 && memcmp(got_kp, c.kp.data(), 4 * (size_t)cnt) != 0);
            if (ds_bad || kp_bad) {
                ++bad;
                if (bad <= 10) {
                    printf("MISMATCH at iteration %ld: %zu scores differ (first index %d: got %.9g expected %.9g); keypoints %d expected %d%s\n",
                           iters, ds_bad, first, first >= 0 ? got_scores[first] : 0.f, first >= 0 ? c.scores[first] : 0.f, cnt, c.n_kp,
                           kp_bad ? " (list differs)" : "");
                    char name[256];
                    snprintf(name, sizeof(name), "gpurun_out/hammer_mismatch_%ld.bin", iters);
                    if (FILE *f = fopen(name, "wb")) {
                        fwrite(&cnt, 4, 1, f);
                        fwrite(got_scores, 4, (size_t)n, f);
                        fwrite(got_kp, 4, (size_t)(cnt > 0 && cnt <= n ? cnt : 0), f);
                        fclose(f);
                    }
                    fflush(stdout);
                }
            }
            if (syncdev) HIPCHECK(hipDeviceSynchronize());
        }
    }
    const double secs = elapsed();
    g_stop = true;
    if (loader.joinable()) loader.join();
    printf("hammer: %ld iterations in %.1f s (%.1f us each), mode %s%s%s%s, load calls %ld, mismatches %ld\n", iters, secs,
           1e6 * secs / (double)(iters ? iters : 1), device ? "device" : "host", load ? " +load" : "", fresh ? " +fresh" : newhandle ? " +newhandle" : "",
           syncdev ? " +syncdev" : "", g_load_calls.load(), bad);
    kpl_destroy(h);
    return bad ? 1 : 0;
}
