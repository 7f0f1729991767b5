// api.cpp -- the C-ABI of libkpl (include/kpl.h): handle, parameter/forest state, and the
// orchestration of the gfx950 kernels.  Host code only; every compute step is a HIP kernel in
// kernels.hip -- there is no CPU fallback.
//
// Reference counterparts: pcl::keypoints::KeypointLearningDetector
// (/root/reference/include/KeypointLearning.h:55-206, include/impl/KeypointLearning.hpp).
#include "../../include/kpl.h"
#ifdef KPL_TEST_HOOKS
#include "../../include/kpl_debug.h"
#endif

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "forest.h"
#include "kernels.h"
#include "organized_normals.h"
#include "soft_pair.h"

using namespace kpl;

// -DKPL_TRACE_HOST (scratch builds only, tools/first_call.py --trace): host-side time stamps of the steps of a call on stderr
#ifdef KPL_TRACE_HOST
#include <chrono>
static void kpl_trace(const char *what) {
    static auto t0 = std::chrono::steady_clock::now();
    static auto last = t0;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[kpl %9.3f ms +%8.3f] %s\n", std::chrono::duration<double, std::milli>(now - t0).count(),
            std::chrono::duration<double, std::milli>(now - last).count(), what);
    last = now;
}
#define KPL_TRACE(what) kpl_trace(what)
#else
#define KPL_TRACE(what) ((void)0)
#endif

namespace {


struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool pooled = false;          // allocated by hipMallocAsync (freed in stream order), else by hipMalloc
    // Grow-only scratch; the content is not kept.  Without a stream: hipFree + hipMalloc -- hipFree waits for EVERY stream of
    // the device (entry points that block anyway: host-buffer calls, set-up).
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        release();
        size_t want = (bytes + bytes / 4 + 256 + 3) & ~(size_t)3;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    // With the stream the next kernels are launched on: the new array is allocated -- and cleared, if asked -- IN THE ORDER OF
    // THAT STREAM (hipMallocAsync, hipMemsetAsync), the old one is freed in that order too (hipFreeAsync: after the kernels of
    // earlier calls that still use it).  Nothing waits on the host, no other stream or handle of the process is held up; an
    // array that came from hipMalloc is parked and freed with the handle.
    hipError_t ensure(size_t bytes, hipStream_t st, std::vector<void *> &parked, bool zero = false) {
        if (bytes <= cap) return hipSuccess;
        const size_t want = (bytes + bytes / 4 + 256 + 3) & ~(size_t)3;
        void *np = nullptr;
        hipError_t e = hipMallocAsync(&np, want, st);
        if (e != hipSuccess) return e;
        if (zero) {
            // cleared by a kernel of this library, in the order of the stream like every other kernel of the call -- not by
            // hipMemsetAsync (see setup_host_path: the one unexplained event of round 6 had a runtime clear of pool memory in it)
            kpl::launch_zero(np, want & ~(size_t)3, st);
            e = hipGetLastError();
            if (e != hipSuccess) {
                (void)hipFreeAsync(np, st);
                return e;
            }
        }
        if (p) {
            if (pooled) (void)hipFreeAsync(p, st);
            else parked.push_back(p);
        }
        p = np;
        cap = want;
        pooled = true;
        return hipSuccess;
    }
    void release() {
        if (p) {
            if (pooled) {       // hipFree waits for every stream of the device before it releases; hipFreeAsync does not
                (void)hipDeviceSynchronize();
                (void)hipFreeAsync(p, nullptr);
            } else {
                (void)hipFree(p);
            }
        }
        p = nullptr;
        cap = 0;
        pooled = false;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace

struct kpl_detector {
    int device = 0;
    std::string err;
    kpl_params prm;

    bool has_forest = false;
    ForestModel model;
    FlatForest flat;
    DevBuf d_nodes;

    // bound view
    const char *d_xyz = nullptr, *d_nrm = nullptr;
    size_t xs = 0, ns = 0;
    int n = 0;
    bool bound = false;
    bool index_valid = false;
    bool pos_of_valid = false;    // the index was built with pos_of[]
    double index_radius = 0.0;
    bool has_origin = false;      // kpl_set_grid_origin
    // how the feature kernels walk the neighborhoods (never WHAT they compute): kpl_set_feature_walk, or -- automatic -- the
    // mean K_f the handle's previous calls measured (DevState::kf_sum / kf_points, read back in sync_status)
    int walk_forced = KPL_WALK_AUTO, lanes_forced = 0;
    int last_walk = -1, last_lanes = 0, last_words = 0;        // what the last launch took (kpl_timing)
    int scan_poll_limit = 1 << 22;   // per handle; only a library built with -DKPL_TEST_HOOKS can change it (include/kpl_debug.h)
    double kf_hint = -1.0;        // mean neighbors per point of the calls before the last status read; < 0: not known
    double kf_hint_radius = 0.0;  // ... measured at this feature radius
    int kf_hint_n = 0;            // ... on a view of this many points
    unsigned long long kf_seen_sum = 0, kf_seen_points = 0;     // DevState::kf_sum / kf_points at that read
    int lcap_hint = 0;            // sorted-search mode: keys per point the lists of the register-sort kernel need (0: not known = 128),
    double lcap_hint_radius = 0.0;    // ... for this feature radius
    bool all_large_hint = false;  // sorted-search mode: the last call at that radius listed a quarter of the view's points or more for the collect / add kernels
    bool all_huge_hint = false;   // ... and stored more than 1 024 keys per listed point: all of them for the workgroup-per-point kernel
    int all_large_n = 0;
    double launched_radius = 0.0; // feature radius / points of the last scoring launch (what the next read-back describes)
    int launched_n = 0;
    int launched_all_large = 0;        // ... and FeatDesc::all_large it ran with
    int last_lcap = 0;            // sorted-search mode: list capacity / all_large of the last launch (kpl_get_last_launch)
    float origin[3] = {0.0f, 0.0f, 0.0f};

    DevBuf stage_xyz, stage_nrm, stage_idx, stage_feat;
    DevBuf dstate, cid, btable, cell_start, tmp_idx, scan_tmp, pts, nrm, pos_of;
    DevBuf score_sorted, flags, prefix, stats, out_scores, out_kp, out_count, cand_list, cand_count;
    DevBuf draw_list, draw_count, skip, feat, scan_state;
    DevBuf large_list, seg_start, seg_len, sort_keys;      // sorted-search mode, large neighborhoods (kernels.hip)
    DevBuf words, wseg_start, wseg_len;                    // sorted order through the word lists (sorted_words_kernel): accept words of every point
    bool launched_words = false;  // the last scoring launch was such a one
    int launched_words_lcap = 0;  // ... with lists of so many positions per point (256 / 512)
    bool launched_sorted = false; // the last scoring launch was in sorted order
    int tie_ban = 0;              // sorted order: launches left that list every point without trying the register / word kernels -- the
                                  // last launch that tried them handed most points on although their lists held them (equal distances:
                                  // the stand-ins of sort_position_lists do not order those, the wave / workgroup kernels' 64-bit keys do)
    double kf_estimate = -1.0;    // sorted order, first host call: neighbors per point estimated off the bounding box (estimate_neighborhood)
    double kf_estimate_radius = 0.0;
    int kf_estimate_n = 0;
    double words_mean_keys = -1.0;    // sorted order: keys per listed point of the last call that listed (nearly) every point; < 0: not known
    DevBuf org_scratch;           // kpl_estimate_normals_organized: change map, distance map, integral image
    int cells_cap = 0;            // capacity (cells) of cell_start
    DevState *h_state = nullptr;  // pinned copy of the device state (status read-back)
    int *h_count = nullptr;       // pinned
    hipStream_t stream = nullptr; // the host-buffer entry points enqueue here
    // kpl_host_staging: pinned host buffers the caller fills, uploaded by DMA on a stream of their own
    void *hs_xyz = nullptr, *hs_nrm = nullptr;
    size_t hs_xyz_cap = 0, hs_nrm_cap = 0, hs_xs = 0, hs_ns = 0;
    int hs_n = -1;
    hipStream_t copy_stream = nullptr;
    const void *pending_nrm = nullptr;   // host normals of the view uploaded last, still to be copied (upload_view with defer_normals)
    size_t pending_nrm_bytes = 0;
    hipEvent_t ev_xyz = nullptr, ev_nrm = nullptr;
    DevBuf out_kp_score;
    std::vector<void *> parked;   // hipMalloc'ed arrays that stream-ordered growth replaced: freed with the handle
    void *h_res = nullptr;        // pinned landing zone of the keypoint lists (host-buffer entry points)
    size_t h_res_cap = 0;
    // What the HOST entry points need and the device entry points do not -- the handle's two streams (8-9 ms each to create
    // on this runtime: a hardware queue), the pinned landing buffer, and the first use of the device-to-pinned copy path (the
    // first 128 KiB copy of a handle's stream into pinned memory took 7-8 ms; tools/probes/host_cost_probe.cpp,
    // profiles/r06_first_call.jsonl) -- is set up by a thread that kpl_create starts and the
    // first host entry point joins (streams_ready): a drop-in TestDetector run loads its forest and reads its cloud in the
    // meantime, and its ONE compute() no longer pays 17 of its 18 ms for set-up.  Until the join the thread owns
    // stream / copy_stream / ev_xyz / ev_nrm / h_res* / out_kp, nothing else.
    // the stream of the handle's last enqueuing call: the scratch -- and its stream-ordered growth -- belongs to the handle, so a
    // call that arrives on ANOTHER stream first makes that stream wait for what the earlier one has queued (enter_stream)
    hipStream_t last_st = nullptr;
    bool has_last_st = false;
    hipEvent_t ev_last = nullptr;
    std::thread setup;
    bool setup_pending = false;
    hipError_t setup_err = hipSuccess;

    // optional per-phase event timing (kpl_enable_timing)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;   // created lazily, reused
    size_t ev_used = 0;
    struct Span { int phase; size_t a, b; };
    std::vector<Span> spans;
};

namespace {

int fail(kpl_detector *h, int code, const char *fmt, ...) {
    if (h) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        h->err = buf;
    }
    return code;
}

#define KPL_HIP(h, call)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(h, KPL_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

constexpr size_t kMaxSpans = 4 * 4096;

// records an event on `st`; returns its pool slot or SIZE_MAX when timing is off / full
size_t mark(kpl_detector *h, hipStream_t st) {
    if (!h->timing || h->spans.size() >= kMaxSpans) return SIZE_MAX;
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return SIZE_MAX;
        h->ev_pool.push_back(e);
    }
    const size_t slot = h->ev_used++;
    if (hipEventRecord(h->ev_pool[slot], st) != hipSuccess) return SIZE_MAX;
    return slot;
}

void span(kpl_detector *h, int phase, size_t a, size_t b) {
    if (a != SIZE_MAX && b != SIZE_MAX) h->spans.push_back({phase, a, b});
}

int use_device(kpl_detector *h) {
    KPL_HIP(h, hipSetDevice(h->device));
    return KPL_OK;
}

// Called by every enqueuing entry point with the stream it is about to use.  One handle per stream is the intended use and costs
// nothing here; a handle that moves to another stream (a device entry point on the caller's stream, then a host entry point on
// the handle's own, or two caller streams) gets the ordering its scratch needs without a host wait: an event recorded on the
// earlier stream now -- behind everything the handle has queued there -- and a wait for it on the new one.  The earlier stream
// must still exist (include/kpl.h).
int enter_stream(kpl_detector *h, hipStream_t st) {
    if (h->has_last_st && h->last_st != st) {
        if (!h->ev_last) KPL_HIP(h, hipEventCreateWithFlags(&h->ev_last, hipEventDisableTiming));
        KPL_HIP(h, hipEventRecord(h->ev_last, h->last_st));
        KPL_HIP(h, hipStreamWaitEvent(st, h->ev_last, 0));
    }
    h->last_st = st;
    h->has_last_st = true;
    return KPL_OK;
}

FeatDesc make_feat(const kpl_params &p) {
    FeatDesc f;
    f.A = p.n_annulus;
    f.B = p.n_bins;
    f.F = p.n_annulus * p.n_bins;
    f.support = (float)p.radius_search;       // double search_radius_ -> float `support`
    const PairConsts ca = annulus_consts(f.A, f.support), cb = bin_consts(f.B);      // soft_pair.h (cpp:43,52 / :75,83)
    f.A1f = ca.nm1;
    f.B1f = cb.nm1;
    f.ann_dim = ca.dim;
    f.ann_half = ca.half;
    f.ann_rdim = ca.rdim;                     // correctly rounded reciprocals (float division)
    f.bin_dim = cb.dim;
    f.bin_half = cb.half;
    f.bin_rdim = cb.rdim;
    f.r2 = (float)(p.radius_search * p.radius_search);
    f.rr = (float)(p.radius_search * (1.0 + 1.0 / 1024.0));
    f.sorted = p.neighbor_order == KPL_NEIGHBORS_SORTED ? 1 : 0;
    f.walk = 0;
    f.lanes = 2;
    f.words = 0;
    f.lcap = 0;
    f.all_large = 0;
    return f;
}

// How the canonical order is walked for this launch.  Both walks and both lane counts give the same bits; this only picks
// the fastest for the neighborhood size, which the host learns from the handle's own earlier calls: the feature kernels
// add up K_f over a sample of their waves, sync_status reads the sums back.  Measured on MI355X (tools/walk_sweep.py,
// profiles/r05_walk_sweep.jsonl): search and drain alternating in one kernel, two lanes per point, is fastest up to ~300
// neighbors per point (63 k points: 0.127 against 0.162 ms at K_f = 194; 500 k points: 0.51 against 0.70); from ~400 on the
// two-pass walk with four lanes per point is (63 k points: 0.256 against 0.305 ms at K_f = 388, 0.889 against 1.348 at
// 1 900; 500 k points: 1.11 against 1.12 at 376, 4.96 against 5.51 at 1 690; cheff001 at the reference's default radius,
// K_f = 2 293: 1.09 against 1.78 ms) -- one drain per point instead of one per 24 accept words of the fullest list of the wave.
constexpr double kTwoPassFromKf = 400.0;
// mean neighbors per point up to which 12 / 16 / 20 accept words are used (one 500 k-point view, feature stage in ms at 12 / 16 /
// 20 / 24 words -- K_f = 48: 0.166 / 0.173 / 0.172 / 0.180; 95: 0.260 / 0.257 / 0.257 / 0.271; 124: 0.342 / 0.313 / 0.312 / 0.327;
// 157: 0.453 / 0.406 / 0.376 / 0.393; 192: 0.546 / 0.529 / 0.479 / 0.466; profiles/r05_accept_words.jsonl)
constexpr double kWords12BelowKf = 80.0, kWords16BelowKf = 140.0, kWords20BelowKf = 175.0;
// the handle's measured (or estimated) neighborhood size describes THIS view: same feature radius, size within 25 %
bool kf_hint_fits(const kpl_detector *h, int n) {
    return h->kf_hint >= 0.0 && h->kf_hint_radius == h->prm.radius_search && h->kf_hint_n > 0 &&
           (long long)n * 4 >= (long long)h->kf_hint_n * 3 && (long long)n * 3 <= (long long)h->kf_hint_n * 4;
}
// how many accept words a point collects between two drains in the one-kernel walk (kernels.hip accept_words: small
// neighborhoods run 8-10 % faster with short lists, large ones 13-17 % slower); 0 = the default of 24
int words_for(const kpl_detector *h) {
    if (!kf_hint_fits(h, h->n)) return 0;
    return h->kf_hint <= 80.0 ? 12 : h->kf_hint <= 140.0 ? 16 : h->kf_hint <= 175.0 ? 20 : 0;
}
void choose_walk(const kpl_detector *h, FeatDesc &f) {
    f.walk = 0;
    f.lanes = 2;
    const bool hint_fits = kf_hint_fits(h, h->n);
    if (h->walk_forced != KPL_WALK_AUTO) {
        f.walk = h->walk_forced == KPL_WALK_TWO_PASS ? 1 : 0;
        f.lanes = h->lanes_forced == 4 ? 4 : 2;
    } else if (hint_fits && h->kf_hint >= kTwoPassFromKf) {
        f.walk = 1;
        f.lanes = 4;
    }
    if (f.walk == 0) f.words = words_for(h);        // (thresholds: kWords12BelowKf / 16 / 20 above)
}
// sorted order through the word lists: mean neighbors per point up to which a view enters / stays, and up to which 256 positions
// per point are the better choice (beyond: 512)
// (what decides about entering is the keys an all_large launch STORED per listed point, chunk tails included: 400 at a true
// mean of 375, 482 at 430, 490 at 460, 501 at 490 -- what decides about staying is the sampled mean itself)
constexpr double kWordsEnterBelow = 495.0, kWordsStayBelow = 480.0, kWords256BelowKf = 225.0;
static_assert(kWords12BelowKf == 80.0 && kWords16BelowKf == 140.0 && kWords20BelowKf == 175.0, "words_for() spells these out");

NmsDesc make_nms(const kpl_params &p) {
    NmsDesc d;
    d.r2 = (float)(p.non_max_radius * p.non_max_radius);
    d.rr = (float)(p.non_max_radius * (1.0 + 1.0 / 1024.0));
    d.thr = p.prediction_th;
    d.non_maxima = p.non_maxima;
    d.draws_remove = p.non_maxima && p.non_maxima_draws_remove;
    d.draws_thr = p.non_maxima_draws_threshold;
    d.scan_poll_limit = 1 << 22;
    return d;
}

int check_params_for_compute(kpl_detector *h, bool need_forest) {
    const kpl_params &p = h->prm;
    if (!(p.radius_search > 0.0) || !std::isfinite(p.radius_search))
        return fail(h, KPL_ERR_INVALID_ARG,
                    "radius_search must be > 0 (k-search mode is not supported: the reference "
                    "passes search_radius_ == 0 as the annulus support)");
    if (p.n_annulus < 1 || p.n_bins < 1) return fail(h, KPL_ERR_INVALID_ARG, "n_annulus and n_bins must be >= 1");
    if ((int64_t)p.n_annulus * p.n_bins > 255)
        return fail(h, KPL_ERR_UNSUPPORTED, "n_annulus * n_bins must be <= 255");
    if (!(p.non_max_radius >= 0.0) || !std::isfinite(p.non_max_radius))
        return fail(h, KPL_ERR_INVALID_ARG, "non_max_radius must be >= 0");
    if (p.neighbor_order != KPL_NEIGHBORS_CANONICAL && p.neighbor_order != KPL_NEIGHBORS_SORTED)
        return fail(h, KPL_ERR_INVALID_ARG, "neighbor_order must be KPL_NEIGHBORS_CANONICAL or KPL_NEIGHBORS_SORTED");
    if (need_forest) {
        if (!h->has_forest) return fail(h, KPL_ERR_NO_FOREST, "no forest loaded");
        if (p.n_annulus * p.n_bins != h->flat.var_count)
            return fail(h, KPL_ERR_VAR_COUNT, "n_annulus*n_bins = %d but the forest has var_count = %d",
                        p.n_annulus * p.n_bins, h->flat.var_count);
    }
    return KPL_OK;
}

int install_forest(kpl_detector *h, ForestModel &&m) {
    FlatForest flat;
    std::string err;
    if (!flatten_forest(m, flat, err)) return fail(h, KPL_ERR_FOREST_PARSE, "forest: %s", err.c_str());
    int rc = use_device(h);
    if (rc) return rc;
    // the new device copy is complete before the handle sees any of it: a failure on the way
    // leaves the previous forest (host and device side) in place
    DevBuf nodes;                           // (the root of tree t is slot t: no table of roots)
    hipError_t e = nodes.ensure(sizeof(FlatNode) * flat.nodes.size());
    if (e == hipSuccess) e = hipMemcpy(nodes.p, flat.nodes.data(), sizeof(FlatNode) * flat.nodes.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        nodes.release();
        return fail(h, KPL_ERR_DEVICE, "forest upload failed: %s", hipGetErrorString(e));
    }
    // REPLACING a forest is the one place left that waits for the whole device: no kernel of an earlier call may still walk
    // the old nodes, and the loaders have no stream to order the release in (a set-up call; the first forest of a handle
    // replaces nothing and waits for nothing)
    if (h->d_nodes.p) KPL_HIP(h, hipDeviceSynchronize());
    h->d_nodes.release();
    h->d_nodes = nodes;
    h->model = std::move(m);
    h->flat = std::move(flat);
    h->has_forest = true;
    return KPL_OK;
}

// (re)allocates the cell table for `cap` cells, cleared, in the order of `st` (the stream the index kernels that read it are
// launched on next).  Rounds 3 and 4 cleared it with hipMemset on the null stream, which returns BEFORE the clear is done and
// is not ordered against a non-blocking stream: cell_sort_store_kernel wrote cell_start[] while the clear was still passing
// over it and lost its writes (every score NaN, status OK: tools/repro_table_growth.py, profiles/r04_notes.md section 1).
// Round 4 fenced that with hipDeviceSynchronize on both sides -- correct, but a barrier for every stream of the process.  A
// clear ON the launch stream is ordered by the stream itself, and so is the release of the old table (DevBuf::ensure).
int ensure_cells(kpl_detector *h, int64_t cap, hipStream_t st) {
    if (cap <= h->cells_cap) return KPL_OK;
    if (cap > kMaxGridCells) cap = kMaxGridCells;
    KPL_HIP(h, h->cell_start.ensure(sizeof(int) * ((size_t)cap + 2), st, h->parked, true));
    h->cells_cap = (int)cap;
    return KPL_OK;
}

// the tables of the index build for a view of n points, in the order of `st` (prepare_index, kpl_reserve)
int ensure_index_tables(kpl_detector *h, int n, hipStream_t st) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    if (h->cells_cap == 0) {
        int rc = ensure_cells(h, (int64_t)8 * n + 65536, st);
        if (rc) return rc;
    }
    KPL_HIP(h, h->cid.ensure(sizeof(int) * nn, st, h->parked));
    KPL_HIP(h, h->tmp_idx.ensure(2 * sizeof(float4) * nn, st, h->parked));
    KPL_HIP(h, h->btable.ensure(sizeof(int) * btable_ints(n), st, h->parked));
    const size_t scan_len = (size_t)(h->cells_cap > n ? h->cells_cap : n) + 1;
    KPL_HIP(h, h->scan_tmp.ensure(sizeof(int) * (scan_len / 4096 + 4), st, h->parked));
    KPL_HIP(h, h->pts.ensure(pts_bytes(n), st, h->parked));            // incl. the tail the search steps read past the last point
    KPL_HIP(h, h->nrm.ensure(sizeof(float4) * nn, st, h->parked));
    KPL_HIP(h, h->pos_of.ensure(sizeof(int) * nn, st, h->parked));
    return KPL_OK;
}

// the scratch of the scoring / NMS half that depends on the view size and the parameters only (prepare_detect, kpl_reserve);
// every array in the order of the launch stream, the ones the kernels expect zeroed cleared in that order too (flags /
// cand.count / skip are kept zero by the compaction from then on; scan_state: tag 0 = "never written")
int ensure_detect_scratch(kpl_detector *h, int n, hipStream_t st) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    KPL_HIP(h, h->score_sorted.ensure(sizeof(float) * nn, st, h->parked));
    KPL_HIP(h, h->feat.ensure(feat_bytes(n, h->prm.n_annulus * h->prm.n_bins), st, h->parked));
    KPL_HIP(h, h->flags.ensure(sizeof(int) * (nn + 1), st, h->parked, true));
    KPL_HIP(h, h->cand_count.ensure(sizeof(int), st, h->parked, true));
    KPL_HIP(h, h->cand_list.ensure(sizeof(int) * nn, st, h->parked));
    KPL_HIP(h, h->prefix.ensure(sizeof(int) * (nn + 2), st, h->parked));
    KPL_HIP(h, h->scan_state.ensure(scan_state_bytes(n), st, h->parked, true));
    if (h->prm.non_maxima && h->prm.non_maxima_draws_remove) {
        KPL_HIP(h, h->skip.ensure(sizeof(int) * nn, st, h->parked, true));
        KPL_HIP(h, h->draw_list.ensure(sizeof(int) * (nn * (2 + kDrawAdj) + 8), st, h->parked));     // list, adjacency counts, adjacency (kernels.hip draw_adj_offset)
        KPL_HIP(h, h->draw_count.ensure(sizeof(int), st, h->parked));
    }
    return KPL_OK;
}

// Index stage ("initCompute"), fully asynchronous: bounding box -> grid descriptor (on the device)
// -> cell ids + counts -> scan -> scatter -> rank/store.  The host does not learn the grid size;
// a view whose grid does not fit the current cell tables sets DevState::status (kpl_sync_status).
// prepare_index checks, allocates and fills the index half of the view descriptor.
int prepare_index(kpl_detector *h, bool auto_cell, ViewDev &v, hipStream_t st, double cell = 0.0) {
    int rc = (auto_cell || cell > 0.0) ? KPL_OK : check_params_for_compute(h, false);
    if (rc) return rc;
    if (!h->bound) return fail(h, KPL_ERR_NO_CLOUD, "no cloud bound");
    if (h->n >= (1 << 28)) return fail(h, KPL_ERR_UNSUPPORTED, "more than 2^28 - 1 points per view");
    rc = use_device(h);
    if (!rc) rc = enter_stream(h, st);
    if (rc) return rc;
    const int n = h->n;
    rc = ensure_index_tables(h, n, st);
    if (rc) return rc;
    v.xyz = h->d_xyz;
    v.nrmsrc = h->d_nrm;
    v.xs = (unsigned)h->xs;
    v.ns = (unsigned)h->ns;
    v.n = n;
    v.ds = h->dstate.as<DevState>();
    v.cells_cap = h->cells_cap;
    v.cell = auto_cell ? 0.0f : (float)(cell > 0.0 ? cell : h->prm.radius_search);
    v.has_origin = (h->has_origin && !auto_cell) ? 1 : 0;
    for (int k = 0; k < 3; ++k) v.origin[k] = h->origin[k];
    v.cid = h->cid.as<int>();
    v.btable = h->btable.as<int>();
    v.btotal = h->btable.as<int>() + (btable_ints(n) - 2 * (kBuckets + 1));
    v.bstart = v.btotal + (kBuckets + 1);
    v.cell_start = h->cell_start.as<int>();
    v.rec = h->tmp_idx.as<float4>();
    v.want_pos_of = 1;
    v.scan_tmp = h->scan_tmp.as<int>();
    v.pts = h->pts.as<float4>();
    v.nrm = h->nrm.as<float4>();
    v.pos_of = h->pos_of.as<int>();
    return KPL_OK;
}

bool index_is_current(const kpl_detector *h) { return h->index_valid && h->index_radius == h->prm.radius_search; }

void index_was_built(kpl_detector *h, bool auto_cell, bool with_pos_of = true) {
    h->index_valid = !auto_cell;
    h->index_radius = h->prm.radius_search;
    h->pos_of_valid = with_pos_of;
}

int build_index(kpl_detector *h, hipStream_t st, bool auto_cell = false) {
    Batch b{};
    b.nviews = 1;
    int rc = prepare_index(h, auto_cell, b.view[0], st);
    if (rc) return rc;
    const size_t ev0 = mark(h, st);
    launch_index(b, st);
    span(h, 0, ev0, mark(h, st));
    KPL_HIP(h, hipGetLastError());
    index_was_built(h, auto_cell);
    return KPL_OK;
}

// waits for `st`, reads the device status of the last index build and turns it into a status
// code; grows the cell tables when they were too small so that a retry succeeds
int sync_status(kpl_detector *h, hipStream_t st) {
    KPL_HIP(h, hipMemcpyAsync(h->h_state, h->dstate.p, sizeof(DevState), hipMemcpyDeviceToHost, st));
    KPL_HIP(h, hipStreamSynchronize(st));
    if (h->h_state->kf_points > h->kf_seen_points) {        // what the scoring launches since the last read measured
        // (only a call that ran to its end describes the view: the search kernel of a call that fails with "word list too
        // small" samples waves that walked nothing, and a hint pulled down by those zeros flips the walk for one call)
        if (h->h_state->status == kStatusOk) {
            h->kf_hint = (double)(h->h_state->kf_sum - h->kf_seen_sum) / (double)(h->h_state->kf_points - h->kf_seen_points);
            h->kf_hint_radius = h->launched_radius;
            h->kf_hint_n = h->launched_n;
        }
        h->kf_seen_sum = h->h_state->kf_sum;
        h->kf_seen_points = h->h_state->kf_points;
    }
    const bool scan_failed = h->h_state->scan_fail != 0;
    if (scan_failed) {
        // a workgroup of the compaction's single-pass scan never saw its predecessors publish (kernels.hip,
        // compact_scan_kernel): the keypoint list of that call is invalid.  The flag is cleared HERE, before any return of
        // this function -- whichever status the call ends with, the next call starts clean
        KPL_HIP(h, hipMemsetAsync((char *)h->dstate.p + offsetof(DevState, scan_fail), 0, sizeof(int), st));
        KPL_HIP(h, hipStreamSynchronize(st));
    }
    if (h->h_state->kf_max > 0) {
        // sorted-search mode: the longest neighborhood the register-sort kernel scored (129 = a list ran full): the capacity
        // of the next launch's lists, a few keys above it in steps of 8 (never below what was seen: no point is deferred
        // that was not deferred before); the device's maximum starts over
        const int seen = h->h_state->kf_max;
        h->lcap_hint = seen > 124 ? 128 : ((seen + 4 + 7) / 8) * 8;
        h->lcap_hint_radius = h->launched_radius;
        h->all_large_hint = h->launched_n > 0 && (long long)h->h_state->large_seen * 4 >= (long long)h->launched_n;
        h->all_large_n = h->launched_n;
        KPL_HIP(h, hipMemsetAsync((char *)h->dstate.p + offsetof(DevState, kf_max), 0, sizeof(int), st));
    }
    if (h->h_state->status == kStatusOk && h->launched_sorted && !h->launched_all_large && h->launched_n > 0 &&
        (long long)h->h_state->large_seen * 2 >= (long long)h->launched_n) {
        const bool lists_held = h->launched_words ? (h->kf_hint > 0.0 && h->kf_hint <= 0.8 * (double)h->launched_words_lcap)
                                                  : (h->h_state->kf_max > 0 && h->h_state->kf_max <= 124);
        if (lists_held) h->tie_ban = 16;
    }
    if (h->h_state->status == kStatusOk && h->h_state->large_seen > 0 && !h->launched_words)     // (a view whose points mostly hold thousands of neighbors: see FeatDesc::all_large == 2)
        h->all_huge_hint = (double)h->h_state->keys_needed > 1024.0 * (double)h->h_state->large_seen;
    // sorted order, what a launch that listed every point (all_large) stored per point: between what the register lists hold and
    // ~200 keys the next launch goes through the word lists (sorted_words_kernel), which from then on measures the mean itself
    if (h->h_state->status == kStatusOk && h->launched_all_large && h->h_state->large_seen > 0)
        h->words_mean_keys = (double)(h->h_state->keys_needed - h->h_state->words_needed) / (double)h->h_state->large_seen;
    if (h->h_state->kf_max <= 0 && h->launched_all_large && h->h_state->status == kStatusOk) {
        // A launch with all_large sends every point to the collect / add kernels, so the register-sort kernel measures nothing
        // (kf_max stays 0) and the hint would never be looked at again: a stream of views of one size at one radius that went
        // from dense to sparse would stay on the slow path for good.  What such a launch does measure is the number of keys it
        // stored (keys_needed: every neighbor of every listed point, chunk tails included) and the points it listed: when
        // the mean is back within what the register lists hold, the next launch tries them again (and measures again).
        const long long listed = h->h_state->large_seen;
        const double mean_keys = listed > 0 ? (double)h->h_state->keys_needed / (double)listed : 0.0;
        if (!(h->lcap_hint > 0 && h->lcap_hint_radius == h->launched_radius)) {
            // the launch listed everything on an ESTIMATE (a first host call, estimate_neighborhood): from here on the handle has
            // measurements like any other -- "every point is listed" for views of this size at this radius, lists of 128 keys
            h->lcap_hint = 128;
            h->lcap_hint_radius = h->launched_radius;
            h->all_large_hint = true;
            h->all_large_n = h->launched_n;
        }
        if (mean_keys < 100.0) {
            h->all_large_hint = false;
            h->lcap_hint = 0;               // (not known: 128 keys per point until the next launch has measured)
        }
    }
    if (h->h_state->status == kStatusOk) h->kf_estimate = -1.0;       // (an estimate serves ONE call: the read above has measured)
    if (h->h_state->status == kStatusGridTooLarge)
        return fail(h, KPL_ERR_GRID_TOO_LARGE, "bounding box / radius needs more than 2^28 grid cells");
    if (h->h_state->status == kStatusBadOrigin)
        return fail(h, KPL_ERR_INVALID_ARG, "the grid origin (kpl_set_grid_origin) exceeds the minimum of the view");
    if (scan_failed && h->h_state->status == kStatusOk) {
        h->index_valid = false;              // (like every other deferred failure: nothing of that call is trusted)
        h->pos_of_valid = false;
        return fail(h, KPL_ERR_INTERNAL, "keypoint compaction: the look-back of the single-pass scan timed out (call again)");
    }
    if (h->h_state->status == kStatusKeyCapacity) {
        // sorted-search mode: the neighbor keys of the points with large neighborhoods did not fit; two-pass walk: the accept
        // words of the view did not (kernels.hip) -- the same array, grown to what the failed call counted
        const unsigned long long need = h->h_state->keys_needed;
        if (need > 0xfffffff0ull) {
            h->kf_hint = -1.0;              // (an automatic choice of the two-pass walk is not repeated: the lanes walk needs no list)
            h->index_valid = false;
            return fail(h, KPL_ERR_CAPACITY, "%llu neighbor keys / accept words in one view (limit 2^32)", need);
        }
        h->index_valid = false;
        if (h->launched_words) {        // sorted order through the word lists: two arrays, each grown to what the call asked of it
            const unsigned long long wneed = h->h_state->words_needed, kneed = need - wneed;
            KPL_HIP(h, h->words.ensure(sizeof(uint2) * (size_t)(wneed + wneed / 16 + 4096), st, h->parked));
            KPL_HIP(h, h->sort_keys.ensure(sizeof(unsigned long long) * (size_t)(kneed + kneed / 16 + 4096), st, h->parked));
        } else {
            KPL_HIP(h, h->sort_keys.ensure(sizeof(unsigned long long) * (size_t)(need + need / 16 + 4096), st, h->parked));
        }
        return fail(h, KPL_ERR_RETRY, "the view needs room for %llu neighbor keys / accept words: array grown, call again", need);
    }
    if (h->h_state->status == kStatusCellCapacity) {
        const int64_t need = h->h_state->ncells_needed;
        h->index_valid = false;
        int rc = ensure_cells(h, need + need / 4 + 1024, st);
        if (rc) return rc;
        return fail(h, KPL_ERR_RETRY, "grid needs %lld cells: tables grown, call again", (long long)need);
    }
    return KPL_OK;
}

// checks, scratch and the scoring / NMS half of the view descriptor (detectKeypoints)
int prepare_detect(kpl_detector *h, float *d_scores, int *d_kp_idx, int kp_cap, int *d_kp_count,
                   StatsDev *d_stats, ViewDev &v, hipStream_t st) {
    int rc = check_params_for_compute(h, true);
    if (rc) return rc;
    if (!h->bound) return fail(h, KPL_ERR_NO_CLOUD, "no cloud bound");
    if (kp_cap < 0 || !d_kp_count || (kp_cap > 0 && !d_kp_idx))
        return fail(h, KPL_ERR_INVALID_ARG, "bad keypoint output buffers");
    rc = use_device(h);
    if (!rc) rc = enter_stream(h, st);
    if (rc) return rc;
    const int n = h->n;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    rc = ensure_detect_scratch(h, n, st);
    if (rc) return rc;
    NmsDesc nd = make_nms(h->prm);
    nd.scan_poll_limit = h->scan_poll_limit;
    v.large_list = nullptr;
    v.sort_keys = nullptr;
    v.seg_start = nullptr;
    v.seg_len = nullptr;
    v.key_cap = 0;
    FeatDesc feat = make_feat(h->prm);
    choose_walk(h, feat);
    if (feat.sorted) {                  // (the walks of the canonical order are not the sorted order's: see words_mode below)
        feat.walk = 0;
        feat.lanes = 2;
        feat.words = 0;
    }
    const bool estimated = feat.sorted && !(h->lcap_hint > 0 && h->lcap_hint_radius == h->prm.radius_search) && h->kf_estimate > 0.0 &&
                           h->kf_estimate_radius == h->prm.radius_search && h->kf_estimate_n == n;
    if (estimated) feat.all_large = h->kf_estimate > 1100.0 ? 2 : 1;       // (a first host call, sorted order: estimate_neighborhood)
    if (feat.sorted && h->lcap_hint > 0 && h->lcap_hint_radius == h->prm.radius_search) {
        feat.lcap = h->lcap_hint;
        feat.all_large = h->all_large_hint && (long long)n * 4 >= (long long)h->all_large_n * 3 && (long long)n * 3 <= (long long)h->all_large_n * 4 ? 1 : 0;
        if (feat.all_large && h->all_huge_hint) feat.all_large = 2;
    }
    if (!feat.sorted && feat.walk == 1) {
        // two-pass walk: the accept words of every point's whole walk (8-byte entries in the array the sorted mode keeps its
        // keys in -- a view is in one mode or the other).  About one word per five neighbors on a surface (32 candidates
        // a word, 4.6 candidates per neighbor) and a third more for the lock step of a wave; a view that needs more fails
        // its first call with KPL_ERR_RETRY and finds the array grown (sync_status).  The hint sizes the list only when it
        // describes this view.  An AUTOMATIC choice is bounded: a list beyond 8 GiB or 2^32 entries (a million points at the
        // reference's default radius would ask for 6 GB), or one the device cannot allocate, sends the launch back to the
        // one-kernel walk, which needs no list -- the same bits, slower for such neighborhoods, but it runs.
        const bool forced = h->walk_forced == KPL_WALK_TWO_PASS;
        const double per_point = (kf_hint_fits(h, n) && h->kf_hint > 0.0 ? h->kf_hint : 1000.0) * 0.3 + 96.0;
        double entries = (double)nn * per_point;
        bool ok = forced || (entries * 8.0 <= 8.0 * 1073741824.0 && entries <= 4.0e9);
        if (entries > 4.2e9) entries = 4.2e9;                                // (forced: entry numbers are 32-bit, the kernels say RETRY / CAPACITY)
        const size_t want = sizeof(unsigned long long) * (size_t)entries;
        if (ok && h->sort_keys.cap < want) {
            const hipError_t e = h->sort_keys.ensure(want, st, h->parked);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                if (forced) return fail(h, KPL_ERR_DEVICE, "word list of the two-pass walk (%zu bytes): %s", want, hipGetErrorString(e));
                ok = false;
            }
        }
        if (!ok) {
            feat.walk = 0;
            feat.lanes = 2;
            feat.words = words_for(h);
        }
    }
    // sorted order, 100 .. ~460 neighbors per point on average: through the word lists (sorted_words_kernel: 256 or 512 positions
    // per point, eight lanes each; points beyond, and points with equal distances, are listed for the wave / workgroup kernels).
    // Entered from what an all_large launch stored per point, kept while the kernel's own sample of the mean stays in range.
    // 8 x 200 k points, feature stage: K_f 190 -> 3.5 ms with 256 positions / 4.6 with 512; 230 -> 5.4 / 5.5; 275 -> 10.2 / 6.6
    // (wave kernel: 9.1); 375 -> 8.3 (11.3); 430 -> 9.2 (12.8); 460 -> 13.4; 490 -> 23.2 (21.4)
    bool words_mode = false;
    if (feat.sorted) {
        const bool sized_alike = h->all_large_n > 0 && (long long)n * 4 >= (long long)h->all_large_n * 3 && (long long)n * 3 <= (long long)h->all_large_n * 4;
        const bool stay = h->launched_words && h->lcap_hint_radius == h->prm.radius_search && sized_alike && kf_hint_fits(h, n) &&
                          h->kf_hint >= 95.0 && h->kf_hint <= kWordsStayBelow;
        if (h->launched_words && !stay && kf_hint_fits(h, n)) h->words_mean_keys = h->kf_hint;      // (the fresher figure decides about coming back)
        const bool enter = feat.all_large == 1 && h->words_mean_keys >= 100.0 && h->words_mean_keys <= kWordsEnterBelow;
        words_mode = (enter || stay) && h->tie_ban == 0;
        if (h->tie_ban > 0) {
            feat.all_large = feat.all_large == 2 ? 2 : 1;
            --h->tie_ban;
        }
    }
    if (words_mode) {
        const double mean = h->launched_words && kf_hint_fits(h, n) ? h->kf_hint : h->words_mean_keys;
        const double entries = (double)nn * (mean * 0.3 + 96.0);
        if (entries * 8.0 <= 8.0 * 1073741824.0 && entries <= 4.0e9) {
            const size_t want = sizeof(uint2) * (size_t)entries;
            if (h->words.cap < want && h->words.ensure(want, st, h->parked) != hipSuccess) {
                (void)hipGetLastError();
                words_mode = false;
            }
        } else {
            words_mode = false;
        }
    }
    if (words_mode) {
        const double mean = h->launched_words && kf_hint_fits(h, n) ? h->kf_hint : h->words_mean_keys;
        feat.walk = 1;
        feat.lanes = 8;
        feat.lcap = mean <= kWords256BelowKf ? 256 : 512;
        feat.all_large = 0;
        KPL_HIP(h, h->wseg_start.ensure(sizeof(unsigned) * nn, st, h->parked));
        KPL_HIP(h, h->wseg_len.ensure(sizeof(int) * nn, st, h->parked));
    }
    h->launched_words = words_mode;
    h->launched_words_lcap = words_mode ? feat.lcap : 0;
    v.words = words_mode ? h->words.as<uint2>() : nullptr;
    v.word_cap = words_mode ? h->words.cap / sizeof(uint2) : 0;
    if (v.word_cap > 0xfffffff0ull) v.word_cap = 0xfffffff0ull;
    v.wseg_start = h->wseg_start.as<unsigned>();
    v.wseg_len = h->wseg_len.as<int>();
    h->last_walk = feat.sorted ? (words_mode ? 1 : -1) : feat.walk;
    h->last_lanes = feat.sorted ? (words_mode ? feat.lanes : 0) : feat.lanes;
    h->last_words = (!feat.sorted && feat.walk == 0) ? (feat.words > 0 ? feat.words : 24) : 0;
    h->last_lcap = feat.sorted ? (feat.lcap > 0 ? feat.lcap : 128) : 0;
    h->launched_all_large = feat.sorted ? feat.all_large : 0;
    h->launched_sorted = feat.sorted != 0;
    if (!feat.sorted && feat.walk == 1) {
        KPL_HIP(h, h->seg_start.ensure(sizeof(unsigned) * nn, st, h->parked));
        KPL_HIP(h, h->seg_len.ensure(sizeof(int) * nn, st, h->parked));
        v.sort_keys = h->sort_keys.as<unsigned long long>();
        v.seg_start = h->seg_start.as<unsigned>();
        v.seg_len = h->seg_len.as<int>();
        v.key_cap = h->sort_keys.cap / sizeof(unsigned long long);
        if (v.key_cap > 0xfffffff0ull) v.key_cap = 0xfffffff0ull;         // entry numbers are 32-bit
    }
    if (h->prm.neighbor_order == KPL_NEIGHBORS_SORTED) {
        // segments of sorted neighbor keys for the points with large neighborhoods: 64 keys per point to begin with; a
        // view that needs more fails its first call with KPL_ERR_RETRY and finds the array grown (sync_status)
        KPL_HIP(h, h->large_list.ensure(sizeof(int) * 2 * nn, st, h->parked));         // all large points + the ones for the workgroup kernel
        KPL_HIP(h, h->seg_start.ensure(sizeof(unsigned) * nn, st, h->parked));
        KPL_HIP(h, h->seg_len.ensure(sizeof(int) * nn, st, h->parked));
        size_t key_bytes = sizeof(unsigned long long) * 64 * nn;
        if (estimated) key_bytes = sizeof(unsigned long long) * (size_t)((double)nn * h->kf_estimate * 1.25);
        if (h->sort_keys.cap < key_bytes) KPL_HIP(h, h->sort_keys.ensure(key_bytes, st, h->parked));
        v.large_list = h->large_list.as<int>();
        v.sort_keys = h->sort_keys.as<unsigned long long>();
        v.seg_start = h->seg_start.as<unsigned>();
        v.seg_len = h->seg_len.as<int>();
        v.key_cap = h->sort_keys.cap / sizeof(unsigned long long);
        if (v.key_cap > 0xfffffffeull) v.key_cap = 0xfffffffeull;         // segment starts are 32-bit
    }
    v.f = feat;
    h->launched_radius = h->prm.radius_search;
    h->launched_n = n;
    v.forest = ForestDev{h->d_nodes.as<uint2>(), h->flat.ntrees, (int)h->flat.nodes.size(), (int)h->flat.ntop,
                         h->flat.order_free ? 1 : 0, h->flat.chain};
    v.nd = nd;
    v.feat = h->feat.as<float>();
    v.score_sorted = h->score_sorted.as<float>();
    v.scores = d_scores;
    v.flags = h->flags.as<int>();
    v.prefix = h->prefix.as<int>();
    v.scan_state = h->scan_state.as<unsigned long long>();
    v.cand = NmsList{h->cand_list.as<int>(), h->cand_count.as<int>()};
    v.draw_list = h->draw_list.as<int>();
    v.draw_count = h->draw_count.as<int>();
    v.skip = h->skip.as<int>();
    v.kp_idx = d_kp_idx;
    v.kp_score = nullptr;
    v.kp_cap = kp_cap;
    v.kp_count = d_kp_count;
    v.stats = d_stats;
    return KPL_OK;
}

// compute() / detectKeypoints of `count` views: one stream, three stages, every kernel launched
// once for the whole batch.  rebuild = always rebuild the index (compute), else only if stale.
int run_batch(kpl_detector *const *handles, int count, float *const *d_scores, int *const *d_kp_idx,
              const int *kp_caps, int *const *d_kp_counts, StatsDev *d_stats, bool rebuild, hipStream_t st,
              float *const *d_kp_scores = nullptr, hipEvent_t normals_ready = nullptr) {
    kpl_detector *h0 = handles[0];
    Batch all{}, idx{}, fix{};
    bool rebuilt[kMaxBatch] = {};
    for (int k = 0; k < count; ++k) {
        kpl_detector *h = handles[k];
        int rc = check_params_for_compute(h, true);
        if (!rc) rc = prepare_index(h, false, all.view[k], st);  // allocates the view's tables ...
        KPL_TRACE("run_batch: index tables ensured");
        if (!rc) rc = prepare_detect(h, d_scores ? d_scores[k] : nullptr, d_kp_idx[k], kp_caps[k], d_kp_counts[k],
                                     d_stats, all.view[k], st);  // ... and its scratch
        if (rc) {
            if (h != h0) fail(h0, rc, "view %d: %s", k, kpl_last_error(h));
            return rc;
        }
        if (d_kp_scores && d_kp_scores[k] && all.view[k].scores) all.view[k].kp_score = d_kp_scores[k];
        // pos_of[] is read by the draws pass only; a view whose index is kept needs it completed if missing
        const bool rebuild_k = rebuild || !index_is_current(h);
        all.view[k].want_pos_of = all.view[k].nd.draws_remove ? 1 : 0;
        if (rebuild_k) idx.view[idx.nviews++] = all.view[k];
        else if (all.view[k].want_pos_of && !h->pos_of_valid) fix.view[fix.nviews++] = all.view[k];
        rebuilt[k] = rebuild_k;
    }
    all.nviews = count;
    KPL_TRACE("run_batch: tables and scratch ensured");
    const size_t ev0 = mark(h0, st);
    if (idx.nviews) {
        launch_index_points(idx, st);
        KPL_TRACE("run_batch: index kernels (points) launched");
        // (host views: the normals arrive on the copy stream while the kernels above run -- from pinned staging buffers
        // their DMA is in flight already, from pageable memory the copy is issued now)
        if (count == 1 && h0->pending_nrm) {
            KPL_HIP(h0, hipMemcpyAsync(h0->stage_nrm.p, h0->pending_nrm, h0->pending_nrm_bytes, hipMemcpyHostToDevice, h0->copy_stream));
            KPL_HIP(h0, hipEventRecord(h0->ev_nrm, h0->copy_stream));
            h0->pending_nrm = nullptr;
            normals_ready = h0->ev_nrm;
        }
        if (normals_ready) KPL_HIP(h0, hipStreamWaitEvent(st, normals_ready, 0));
        launch_index_records(idx, st);
    }
    if (fix.nviews) launch_pos_of(fix, st);
    KPL_TRACE("run_batch: index kernels (records) launched");
    const size_t ev1 = mark(h0, st);
    launch_feature_stage(all, st);
    KPL_TRACE("run_batch: feature stage launched");
    const size_t ev1b = mark(h0, st);
    launch_forest_stage(all, st);
    KPL_TRACE("run_batch: forest stage launched");
    const size_t ev2 = mark(h0, st);
    launch_post(all, st);
    KPL_TRACE("run_batch: NMS + compaction launched");
    const size_t ev3 = mark(h0, st);
    if (idx.nviews) span(h0, 0, ev0, ev1);
    span(h0, 1, ev1, ev1b);
    span(h0, 3, ev1b, ev2);
    span(h0, 2, ev2, ev3);
    KPL_HIP(h0, hipGetLastError());
    KPL_TRACE("run_batch: hipGetLastError");
    for (int k = 0; k < count; ++k) {
        const bool with_map = all.view[k].want_pos_of != 0;
        if (rebuilt[k]) index_was_built(handles[k], false, with_map);
        else if (with_map) handles[k]->pos_of_valid = true;
    }
    return KPL_OK;
}

int detect_on_device(kpl_detector *h, float *d_scores, int *d_kp_idx, int kp_cap, int *d_kp_count,
                     hipStream_t st, StatsDev *d_stats, bool rebuild = false) {
    return run_batch(&h, 1, &d_scores, &d_kp_idx, &kp_cap, &d_kp_count, d_stats, rebuild, st);
}

// pcl::NormalEstimation on the bound view: index on a grid that suits the search (k-search: cell
// from the bounding box, radius search: cell = radius), then one thread per point
int normals_on_device(kpl_detector *h, int k, double radius, const float *viewpoint, void *d_normals, size_t ns,
                      void *d_curv, size_t cs, hipStream_t st) {
    if (!h->bound) return fail(h, KPL_ERR_NO_CLOUD, "no cloud bound");
    if (k > 32) return fail(h, KPL_ERR_UNSUPPORTED, "k_search must be <= 32");
    if (k <= 0 && (!(radius > 0.0) || !std::isfinite(radius)))
        return fail(h, KPL_ERR_INVALID_ARG, "either k_search > 0 or radius_search > 0 is needed");
    if (h->n > 0 && !d_normals) return fail(h, KPL_ERR_INVALID_ARG, "null normals buffer");
    if (ns < 12 || (ns & 3) || (d_curv && (cs < 4 || (cs & 3))))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be multiples of 4 (normals >= 12 bytes)");
    static const float origin[3] = {0.0f, 0.0f, 0.0f};
    Batch b{};
    b.nviews = 1;
    int rc = prepare_index(h, k > 0, b.view[0], st, k > 0 ? 0.0 : radius);
    if (rc) return rc;
    launch_index(b, st);
    const ViewDev &v = b.view[0];
    launch_normals(v.pts, v.cell_start, v.pos_of, v.ds, v.n, k, (float)(radius * radius),
                   (float)(radius * (1.0 + 1.0 / 1024.0)), viewpoint ? viewpoint : origin, (char *)d_normals, ns,
                   (char *)d_curv, cs, st);
    KPL_HIP(h, hipGetLastError());
    h->index_valid = false;        // the index holds whatever sat at the normal pointer before
    return KPL_OK;
}

// A handle that has measured nothing yet for this radius and view size (a drop-in TestDetector run makes ONE call) gets an
// estimate from what the host-buffer entry points have in hand -- the points: a 2.5D view is a surface, so the number of
// neighbors within r is about pi r^2 x points per unit of area, the area about the product of the two largest extents of the
// bounding box (cheff001 at the reference's default radius 20: 2 040 estimated, 2 293 measured; the 200 k-point synthetic
// views at 6 mesh resolutions: 79 / 69).  Only the choice of the walk hangs on it (choose_walk), never a result, and the first
// kpl_sync_status replaces it by the measurement.  One pass over the points, on the first call of a (radius, size) only.
void estimate_neighborhood(kpl_detector *h, const void *xyz, size_t xs, int n) {
    const double r = h->prm.radius_search;
    const bool hint_fits = h->kf_hint >= 0.0 && h->kf_hint_radius == r && h->kf_hint_n > 0 &&
                           (long long)n * 4 >= (long long)h->kf_hint_n * 3 && (long long)n * 3 <= (long long)h->kf_hint_n * 4;
    const bool sorted = h->prm.neighbor_order == KPL_NEIGHBORS_SORTED;
    if (sorted && h->lcap_hint > 0 && h->lcap_hint_radius == r) return;           // (the sorted order's own hints exist already)
    if ((!sorted && hint_fits) || n < 1024 || !(r > 0.0) || (!sorted && h->walk_forced != KPL_WALK_AUTO)) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    long long finite = 0;
    const char *base = static_cast<const char *>(xyz);
    for (int i = 0; i < n; ++i) {
        float p[3];
        memcpy(p, base + (size_t)i * xs, 12);
        if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
        ++finite;
        for (int k = 0; k < 3; ++k) {
            lo[k] = p[k] < lo[k] ? p[k] : lo[k];
            hi[k] = p[k] > hi[k] ? p[k] : hi[k];
        }
    }
    if (finite < 1024) return;
    double e[3] = {(double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2]};
    std::sort(e, e + 3);
    const double area = e[2] * e[1];
    if (!(area > 0.0)) return;
    const double estimate = 3.14159265358979 * r * r * (double)finite / area;
    if (sorted) {
        // the sorted order's first call: with hundreds of neighbors per point the register-sort kernel would search every box only
        // to find its lists full, and the key array (64 keys per point to begin with) would send the call back with RETRY after a
        // whole run -- a one-shot TestDetector --sortedSearch at the reference's default radius paid 15-50 ms for that.  A clear
        // estimate sizes the key array and lists every point at once (prepare_detect); the first status read replaces it
        if (estimate >= 300.0 && (double)n * estimate * 1.25 * 8.0 <= 4.0 * 1073741824.0) {
            h->kf_estimate = estimate;
            h->kf_estimate_radius = r;
            h->kf_estimate_n = n;
        }
        return;
    }
    if (estimate < 1.5 * kTwoPassFromKf) return;   // a rough figure: only a clear case leaves the default before a measurement
    // ... and a VOLUME of points is not a surface (a million points in a cube, r = 1/50 of its edge: 1 260 estimated, 33 real):
    // the two-pass walk sizes its word lists by the hint, so an estimate is not allowed to ask for more than 1 GiB of them
    if ((double)n * (estimate * 0.3 + 96.0) * 8.0 > 1073741824.0) return;
    h->kf_hint = estimate;
    h->kf_hint_radius = r;
    h->kf_hint_n = n;
}

constexpr size_t kLandingEntries = 32768;      // keypoint indices (+ responses) that come back with the count, on speculation

void setup_host_path(kpl_detector *h) {
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_xyz, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_nrm, hipEventDisableTiming);
    const size_t landing = kLandingEntries * (sizeof(int) + sizeof(float));
    if (e == hipSuccess) e = hipHostMalloc(&h->h_res, landing + landing / 2 + 4096, hipHostMallocDefault);
    if (e == hipSuccess) h->h_res_cap = landing + landing / 2 + 4096;
    // first use of the device-to-pinned copy path, on the stream that will use it: the first 128 KiB copy into pinned memory
    // took 7-8 ms inside the first compute().  The source is the handle's own keypoint array (plain hipMalloc, kept), the
    // result is thrown away.  NOT through a stream-ordered scratch block: a block from hipMallocAsync that was cleared with
    // hipMemsetAsync and freed again here made the first compute() of every second handle of a process count no keypoints
    // (flags all zero at the compaction; scores right; the second call right) -- reproduced with the C API alone, not
    // reproduced outside libkpl (tools/probes/memset_order_probe.cpp), gone without the clear or without the pool
    // (profiles/r06_notes.md).  No copy out of pageable memory either: this thread runs next to the caller's loadForest.
    if (e == hipSuccess) e = h->out_kp.ensure(kLandingEntries * sizeof(int));
    if (e == hipSuccess) e = hipMemcpyAsync(h->h_res, h->out_kp.p, kLandingEntries * sizeof(int), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    h->setup_err = e;
}

// every entry point that uses the handle's own streams passes through here first
int streams_ready(kpl_detector *h) {
    if (h->setup_pending) {
        h->setup.join();
        h->setup_pending = false;
        KPL_TRACE("streams_ready: joined the set-up thread");
    }
    if (h->setup_err != hipSuccess)
        return fail(h, KPL_ERR_DEVICE, "set-up of the handle's streams failed: %s", hipGetErrorString(h->setup_err));
    return KPL_OK;
}

// defer_normals: only the points are copied here; the normals are copied by run_batch on the copy stream AFTER it has
// launched the index kernels that read only points -- a copy from pageable memory keeps the calling thread busy while the
// runtime stages it, so those kernels run under it (compute() from pageable PCL records: 0.52 -> 0.47 ms per 200 k points)
int upload_view(kpl_detector *h, const void *xyz, size_t xs, const void *nrm, size_t ns, int n, bool defer_normals = false) {
    if (n < 0 || (n > 0 && (!xyz || !nrm))) return fail(h, KPL_ERR_INVALID_ARG, "null cloud or normals");
    if (xs < 12 || ns < 12 || (xs & 3) || (ns & 3))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be multiples of 4 and >= 12 bytes");
    int rc = use_device(h);
    if (!rc) rc = streams_ready(h);
    if (!rc) rc = enter_stream(h, h->stream);       // (the staging arrays may still be read by a call on another stream)
    if (rc) return rc;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    KPL_TRACE("upload_view: begin");
    KPL_HIP(h, h->stage_xyz.ensure(nn * xs));
    KPL_HIP(h, h->stage_nrm.ensure(nn * ns));
    KPL_TRACE("upload_view: staging arrays");
    if (n > 0) {
        // the last element may be shorter than the stride in the caller's array.  Enqueued on the
        // handle's stream (pageable memory: the runtime stages it, the call returns once it has);
        // everything the host-buffer entry points launch afterwards goes to the same stream
        KPL_HIP(h, hipMemcpyAsync(h->stage_xyz.p, xyz, (size_t)(n - 1) * xs + 12, hipMemcpyHostToDevice, h->stream));
        h->pending_nrm = nullptr;
        if (nrm != xyz) {
            if (defer_normals) {
                h->pending_nrm = nrm;
                h->pending_nrm_bytes = (size_t)(n - 1) * ns + 12;
            } else {
                KPL_HIP(h, hipMemcpyAsync(h->stage_nrm.p, nrm, (size_t)(n - 1) * ns + 12, hipMemcpyHostToDevice, h->stream));
            }
        }
    } else {
        h->pending_nrm = nullptr;
    }
    h->d_xyz = h->stage_xyz.as<char>();
    h->d_nrm = h->stage_nrm.as<char>();
    h->xs = xs;
    h->ns = ns;
    h->n = n;
    h->bound = true;
    h->index_valid = false;
    KPL_TRACE("upload_view: copies issued");
    estimate_neighborhood(h, xyz, xs, n);
    KPL_TRACE("upload_view: neighborhood estimated");
    return KPL_OK;
}

// compute() on a staged view: index build + detect on the handle's stream, then the keypoint list (and,
// if asked for, its scores and / or the scores of all points) into the caller's buffers.  Two waits:
// one for the count, one for the lists of exactly that length.
int detect_staged(kpl_detector *h, int n, float *scores_out, int *kp_idx_out, float *kp_scores_out, int kp_cap,
                  int *kp_count, hipEvent_t normals_ready = nullptr) {
    struct ClearPending {        // normals still to be copied belong to THIS call, however it ends
        kpl_detector *h;
        ~ClearPending() { h->pending_nrm = nullptr; }
    } clear_pending{h};
    const size_t nn = (size_t)(n > 0 ? n : 1);
    const bool want_scores = scores_out != nullptr || kp_scores_out != nullptr;
    if (want_scores) KPL_HIP(h, h->out_scores.ensure(sizeof(float) * nn));
    KPL_HIP(h, h->out_kp.ensure(sizeof(int) * nn));
    if (kp_scores_out) KPL_HIP(h, h->out_kp_score.ensure(sizeof(float) * nn));
    KPL_HIP(h, h->out_count.ensure(sizeof(int)));
    hipStream_t st = h->stream;
    float *d_scores = want_scores ? h->out_scores.as<float>() : nullptr;
    int *d_kp = h->out_kp.as<int>(), *d_count = h->out_count.as<int>();
    float *d_kps = kp_scores_out ? h->out_kp_score.as<float>() : nullptr;
    // The lists come back in ONE round trip: together with the count, the first `spec` entries of the index (and
    // response) list are copied on speculation into the pinned landing buffer -- a 200 k-point view has ~22 k
    // keypoints --; only a longer list costs a second copy and wait.
    int spec = kp_cap < 32768 ? kp_cap : 32768;
    if ((size_t)spec > nn) spec = (int)nn;                   // (the device lists hold n entries)
    const size_t per = sizeof(int) + (kp_scores_out ? sizeof(float) : 0);
    auto ensure_landing = [&](size_t entries) -> int {
        const size_t need = entries * per;
        if (need > h->h_res_cap) {
            if (h->h_res) (void)hipHostFree(h->h_res);
            h->h_res = nullptr;
            h->h_res_cap = 0;
            KPL_HIP(h, hipHostMalloc(&h->h_res, need + need / 2 + 4096, hipHostMallocDefault));
            h->h_res_cap = need + need / 2 + 4096;
        }
        return KPL_OK;
    };
    int rc = ensure_landing((size_t)spec);
    if (rc) return rc;
    KPL_TRACE("detect_staged: output arrays + pinned landing buffer");
    for (int attempt = 0;; ++attempt) {
        rc = run_batch(&h, 1, &d_scores, &d_kp, &n, &d_count, nullptr, true, st, &d_kps, attempt == 0 ? normals_ready : nullptr);
        if (rc) return rc;
        KPL_HIP(h, hipMemcpyAsync(h->h_count, h->out_count.p, sizeof(int), hipMemcpyDeviceToHost, st));
        KPL_TRACE("detect_staged: count copy issued");
        if (spec > 0) {
            KPL_HIP(h, hipMemcpyAsync(h->h_res, h->out_kp.p, sizeof(int) * (size_t)spec, hipMemcpyDeviceToHost, st));
            KPL_TRACE("detect_staged: index list copy issued");
            if (kp_scores_out)
                KPL_HIP(h, hipMemcpyAsync((char *)h->h_res + sizeof(int) * (size_t)spec, h->out_kp_score.p,
                                          sizeof(float) * (size_t)spec, hipMemcpyDeviceToHost, st));
        }
        if (scores_out && n > 0)
            KPL_HIP(h, hipMemcpyAsync(scores_out, h->out_scores.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
        KPL_TRACE("detect_staged: run_batch + copies issued");
        rc = sync_status(h, st);
        KPL_TRACE(rc == KPL_ERR_RETRY ? "detect_staged: synced, RETRY" : "detect_staged: synced");
        if (rc == KPL_ERR_RETRY && attempt < 6) continue;    // cell tables / key segments / word lists were grown: run again
        if (rc) return rc;
        break;
    }
    const int count = h->h_count[0];
    *kp_count = count;
    const int ncopy = count < kp_cap ? count : kp_cap;
    if (ncopy > spec) {                                      // a list longer than the speculative copy: fetch all of it
        rc = ensure_landing((size_t)ncopy);
        if (rc) return rc;
        KPL_HIP(h, hipMemcpyAsync(h->h_res, h->out_kp.p, sizeof(int) * (size_t)ncopy, hipMemcpyDeviceToHost, st));
        if (kp_scores_out)
            KPL_HIP(h, hipMemcpyAsync((char *)h->h_res + sizeof(int) * (size_t)ncopy, h->out_kp_score.p,
                                      sizeof(float) * (size_t)ncopy, hipMemcpyDeviceToHost, st));
        KPL_HIP(h, hipStreamSynchronize(st));
    }
    if (ncopy > 0) {
        const size_t stride = (size_t)(ncopy > spec ? ncopy : spec);
        memcpy(kp_idx_out, h->h_res, sizeof(int) * (size_t)ncopy);
        if (kp_scores_out) memcpy(kp_scores_out, (char *)h->h_res + sizeof(int) * stride, sizeof(float) * (size_t)ncopy);
    }
    if (count > kp_cap) return fail(h, KPL_ERR_CAPACITY, "%d keypoints but capacity %d", count, kp_cap);
    return KPL_OK;
}

// a view staged WITHOUT normals (resolution, normal estimation) must never be mistaken for a bound
// view afterwards, whichever way the entry point returns
struct UnbindOnExit {
    kpl_detector *h;
    ~UnbindOnExit() {
        h->bound = false;
        h->index_valid = false;
    }
};

}  // namespace

// =============================================================================================
extern "C" {

int kpl_version(void) { return KPL_VERSION; }

#ifndef KPL_SOURCE_SHA
#define KPL_SOURCE_SHA "unknown"
#endif
const char *kpl_source_hash(void) { return KPL_SOURCE_SHA; }

const char *kpl_status_string(int s) {
    switch (s) {
        case KPL_OK: return "ok";
        case KPL_ERR_INVALID_ARG: return "invalid argument";
        case KPL_ERR_NO_FOREST: return "no forest loaded";
        case KPL_ERR_FOREST_PARSE: return "forest parse error";
        case KPL_ERR_VAR_COUNT: return "n_annulus*n_bins does not match the forest";
        case KPL_ERR_GRID_TOO_LARGE: return "grid too large";
        case KPL_ERR_CAPACITY: return "output capacity too small";
        case KPL_ERR_DEVICE: return "HIP device error";
        case KPL_ERR_UNSUPPORTED: return "unsupported";
        case KPL_ERR_IO: return "i/o error";
        case KPL_ERR_NO_CLOUD: return "no cloud bound";
        case KPL_ERR_RETRY: return "cell tables grown, call again";
        case KPL_ERR_INTERNAL: return "internal error on the device (result invalid)";
        default: return "unknown status";
    }
}

void kpl_default_params(kpl_params *p) {
    if (!p) return;
    p->n_annulus = 5;                       // KeypointLearning.h:81
    p->n_bins = 10;
    p->radius_search = 0.0;
    p->non_max_radius = 0.0;
    p->prediction_th = 0.5f;
    p->non_maxima = 1;
    p->non_maxima_draws_remove = 1;
    p->non_maxima_draws_threshold = 0.0f;
    p->neighbor_order = KPL_NEIGHBORS_CANONICAL;
}

int kpl_create(kpl_detector **out, int device) {
    if (!out) return KPL_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return KPL_ERR_DEVICE;
    KPL_TRACE("kpl_create: begin");
    kpl_detector *h = new (std::nothrow) kpl_detector();
    if (!h) return KPL_ERR_DEVICE;
    h->device = device;
    kpl_default_params(&h->prm);
    if (hipSetDevice(device) != hipSuccess ||
        hipHostMalloc((void **)&h->h_state, sizeof(DevState), hipHostMallocDefault) != hipSuccess ||
        h->dstate.ensure(sizeof(DevState)) != hipSuccess ||
        hipHostMalloc((void **)&h->h_count, 16 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        kpl_destroy(h);
        return KPL_ERR_DEVICE;
    }
    init_dev_state(h->h_state);
    if (hipMemcpy(h->dstate.p, h->h_state, sizeof(DevState), hipMemcpyHostToDevice) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {          // (the null stream is not ordered against the handle's stream)
        kpl_destroy(h);
        return KPL_ERR_DEVICE;
    }
    preload_code();                  // the code objects of both kernel files, now instead of under the first launch
    try {
        h->setup = std::thread(setup_host_path, h);
        h->setup_pending = true;
    } catch (...) {                  // no thread to be had: set up here
        setup_host_path(h);
    }
    KPL_TRACE("kpl_create: end");
    *out = h;
    return KPL_OK;
}

void kpl_destroy(kpl_detector *h) {
    if (!h) return;
    if (h->setup_pending) h->setup.join();
    h->setup_pending = false;
    (void)hipSetDevice(h->device);
    // kernels of the handle's last calls may still be running, on streams this function knows nothing about: the arrays that
    // came from hipMallocAsync are released with hipFreeAsync, which -- unlike the hipFree of earlier rounds -- waits for nothing
    (void)hipDeviceSynchronize();
    DevBuf *bufs[] = {&h->org_scratch, &h->d_nodes, &h->stage_xyz, &h->stage_nrm, &h->stage_idx, &h->stage_feat,
                      &h->dstate, &h->cid, &h->btable, &h->cell_start, &h->tmp_idx, &h->scan_tmp,
                      &h->pts, &h->nrm, &h->pos_of, &h->score_sorted, &h->flags, &h->prefix, &h->stats,
                      &h->out_scores, &h->out_kp, &h->out_count, &h->cand_list, &h->cand_count,
                      &h->draw_list, &h->draw_count, &h->skip, &h->feat, &h->scan_state,
                      &h->large_list, &h->seg_start, &h->seg_len, &h->sort_keys, &h->words, &h->wseg_start, &h->wseg_len};
    for (DevBuf *b : bufs) b->release();
    for (void *q : h->parked) (void)hipFree(q);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->h_count) (void)hipHostFree(h->h_count);
    if (h->h_res) (void)hipHostFree(h->h_res);
    if (h->hs_xyz) (void)hipHostFree(h->hs_xyz);
    if (h->hs_nrm) (void)hipHostFree(h->hs_nrm);
    if (h->ev_last) (void)hipEventDestroy(h->ev_last);
    if (h->ev_xyz) (void)hipEventDestroy(h->ev_xyz);
    if (h->ev_nrm) (void)hipEventDestroy(h->ev_nrm);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    h->out_kp_score.release();
    delete h;
}

const char *kpl_last_error(const kpl_detector *h) { return h ? h->err.c_str() : "null handle"; }

int kpl_set_params(kpl_detector *h, const kpl_params *p) {
    if (!h || !p) return KPL_ERR_INVALID_ARG;
    if (p->n_annulus < 1 || p->n_bins < 1) return fail(h, KPL_ERR_INVALID_ARG, "n_annulus and n_bins must be >= 1");
    h->prm = *p;
    return KPL_OK;
}

int kpl_set_feature_walk(kpl_detector *h, int walk, int lanes_per_point) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (walk != KPL_WALK_AUTO && walk != KPL_WALK_LANES && walk != KPL_WALK_TWO_PASS)
        return fail(h, KPL_ERR_INVALID_ARG, "walk must be KPL_WALK_AUTO, KPL_WALK_LANES or KPL_WALK_TWO_PASS");
    if (walk != KPL_WALK_AUTO && lanes_per_point != 2 && lanes_per_point != 4)
        return fail(h, KPL_ERR_INVALID_ARG, "lanes_per_point must be 2 or 4");
    h->walk_forced = walk;
    h->lanes_forced = walk == KPL_WALK_AUTO ? 0 : lanes_per_point;
    return KPL_OK;
}

#ifdef KPL_TEST_HOOKS                 // include/kpl_debug.h: not in the shipped library
int kpl_debug_set_scan_poll_limit(kpl_detector *h, int polls) {
    if (!h) return KPL_ERR_INVALID_ARG;
    h->scan_poll_limit = polls;
    return KPL_OK;
}
#endif

int kpl_get_feature_walk(const kpl_detector *h, int *walk, int *lanes_per_point, double *mean_neighbors) {
    if (!h) return KPL_ERR_INVALID_ARG;
    FeatDesc f = make_feat(h->prm);
    choose_walk(h, f);
    if (walk) *walk = f.walk ? KPL_WALK_TWO_PASS : KPL_WALK_LANES;
    if (lanes_per_point) *lanes_per_point = f.lanes;
    if (mean_neighbors) *mean_neighbors = h->kf_hint;
    return KPL_OK;
}

int kpl_get_last_launch(const kpl_detector *h, kpl_launch_info *out) {
    if (!h || !out) return KPL_ERR_INVALID_ARG;
    out->walk = h->last_walk;
    out->lanes_per_point = h->last_lanes;
    out->accept_words = h->last_words;
    out->sorted_list_keys = h->last_lcap;
    out->sorted_all_large = h->launched_all_large;
    return KPL_OK;
}

int kpl_get_params(const kpl_detector *h, kpl_params *p) {
    if (!h || !p) return KPL_ERR_INVALID_ARG;
    *p = h->prm;
    return KPL_OK;
}

int kpl_load_forest_memory(kpl_detector *h, const void *data, size_t len) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (!data || len == 0) return fail(h, KPL_ERR_INVALID_ARG, "empty forest buffer");
    std::string text, err;
    if (!inflate_if_gzip(data, len, text, err)) return fail(h, KPL_ERR_FOREST_PARSE, "%s", err.c_str());
    ForestModel m;
    if (!parse_forest_yaml(text.data(), text.size(), m, err)) return fail(h, KPL_ERR_FOREST_PARSE, "%s", err.c_str());
    return install_forest(h, std::move(m));
}

int kpl_load_forest_file(kpl_detector *h, const char *path) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (!path) return fail(h, KPL_ERR_INVALID_ARG, "null path");
    std::string text, err;
    if (!read_maybe_gzip(path, text, err)) {
        // impl/KeypointLearning.hpp:165-168: "impossible to load random forest"
        return fail(h, err.rfind("cannot open", 0) == 0 ? KPL_ERR_IO : KPL_ERR_FOREST_PARSE, "%s", err.c_str());
    }
    ForestModel m;
    if (!parse_forest_yaml(text.data(), text.size(), m, err)) return fail(h, KPL_ERR_FOREST_PARSE, "%s: %s", path, err.c_str());
    return install_forest(h, std::move(m));
}

int kpl_load_forest_arrays(kpl_detector *h, int ntrees, int nnodes, int var_count, const int *root,
                           const int *var, const float *thr, const int *left, const int *right,
                           const double *value) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (ntrees <= 0 || nnodes <= 0 || !root || !var || !thr || !left || !right || !value)
        return fail(h, KPL_ERR_INVALID_ARG, "null or empty forest arrays");
    ForestModel m;
    m.var_count = var_count;
    m.root.assign(root, root + ntrees);
    m.var.assign(var, var + nnodes);
    m.thr.assign(thr, thr + nnodes);
    m.left.assign(left, left + nnodes);
    m.right.assign(right, right + nnodes);
    m.value.assign(value, value + nnodes);
    return install_forest(h, std::move(m));
}

int kpl_forest_info(const kpl_detector *h, int *ntrees, int *var_count, int64_t *nnodes, int *max_depth) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (!h->has_forest) return KPL_ERR_NO_FOREST;
    if (ntrees) *ntrees = h->flat.ntrees;
    if (var_count) *var_count = h->flat.var_count;
    if (nnodes) *nnodes = h->flat.nnodes;
    if (max_depth) *max_depth = h->flat.max_depth;
    return KPL_OK;
}

static int parse_bytes(const void *data, size_t len, ForestModel &m, char *err, size_t err_cap) {
    std::string text, msg;
    bool ok = data && len && inflate_if_gzip(data, len, text, msg) &&
              parse_forest_yaml(text.data(), text.size(), m, msg);
    if (!ok) {
        if (msg.empty()) msg = "empty forest buffer";
        if (err && err_cap) snprintf(err, err_cap, "%s", msg.c_str());
        return (!data || !len) ? KPL_ERR_INVALID_ARG : KPL_ERR_FOREST_PARSE;
    }
    return KPL_OK;
}

int kpl_forest_inspect(const void *data, size_t len, kpl_forest_summary *out, char *err, size_t err_cap) {
    if (!out) return KPL_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    ForestModel m;
    int rc = parse_bytes(data, len, m, err, err_cap);
    if (rc) return rc;
    FlatForest flat;
    std::string msg;
    if (!flatten_forest(m, flat, msg)) {
        if (err && err_cap) snprintf(err, err_cap, "%s", msg.c_str());
        return KPL_ERR_FOREST_PARSE;
    }
    out->ntrees = flat.ntrees;
    out->var_count = flat.var_count;
    out->nnodes = flat.nnodes;
    out->max_depth = flat.max_depth;
    return KPL_OK;
}

int kpl_forest_export_arrays(const void *data, size_t len, int64_t node_cap, int tree_cap, int *root,
                             int *var, float *thr, int *left, int *right, double *value, char *err,
                             size_t err_cap) {
    ForestModel m;
    int rc = parse_bytes(data, len, m, err, err_cap);
    if (rc) return rc;
    if (m.nnodes() > node_cap || m.ntrees() > tree_cap) {
        if (err && err_cap) snprintf(err, err_cap, "need room for %lld nodes, %d trees", (long long)m.nnodes(), m.ntrees());
        return KPL_ERR_CAPACITY;
    }
    const size_t nn = (size_t)m.nnodes();
    if (root) memcpy(root, m.root.data(), sizeof(int) * m.root.size());
    if (var) memcpy(var, m.var.data(), sizeof(int) * nn);
    if (thr) memcpy(thr, m.thr.data(), sizeof(float) * nn);
    if (left) memcpy(left, m.left.data(), sizeof(int) * nn);
    if (right) memcpy(right, m.right.data(), sizeof(int) * nn);
    if (value) memcpy(value, m.value.data(), sizeof(double) * nn);
    return KPL_OK;
}

int kpl_set_grid_origin(kpl_detector *h, const float *origin) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (origin && !(std::isfinite(origin[0]) && std::isfinite(origin[1]) && std::isfinite(origin[2])))
        return fail(h, KPL_ERR_INVALID_ARG, "grid origin must be finite");
    h->has_origin = origin != nullptr;
    for (int k = 0; k < 3; ++k) h->origin[k] = origin ? origin[k] : 0.0f;
    h->index_valid = false;
    return KPL_OK;
}

int kpl_bind_cloud_device(kpl_detector *h, const void *d_xyz, size_t xyz_stride, const void *d_normals,
                          size_t normals_stride, int n) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!d_xyz || !d_normals))) return fail(h, KPL_ERR_INVALID_ARG, "null cloud or normals");
    if (xyz_stride < 12 || normals_stride < 12 || (xyz_stride & 3) || (normals_stride & 3))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be multiples of 4 and >= 12 bytes");
    h->d_xyz = (const char *)d_xyz;
    h->d_nrm = (const char *)d_normals;
    h->xs = xyz_stride;
    h->ns = normals_stride;
    h->n = n;
    h->bound = true;
    h->index_valid = false;
    return KPL_OK;
}

int kpl_build_index_device(kpl_detector *h, void *stream) {
    if (!h) return KPL_ERR_INVALID_ARG;
    return build_index(h, (hipStream_t)stream);
}

int kpl_detect_device(kpl_detector *h, float *d_scores, int *d_kp_idx, int kp_cap, int *d_kp_count, void *stream) {
    if (!h) return KPL_ERR_INVALID_ARG;
    return detect_on_device(h, d_scores, d_kp_idx, kp_cap, d_kp_count, (hipStream_t)stream, nullptr);
}

int kpl_compute_device(kpl_detector *h, float *d_scores, int *d_kp_idx, int kp_cap, int *d_kp_count, void *stream) {
    if (!h) return KPL_ERR_INVALID_ARG;
    return detect_on_device(h, d_scores, d_kp_idx, kp_cap, d_kp_count, (hipStream_t)stream, nullptr, true);
}

int kpl_compute_batch_device(kpl_detector *const *handles, int count, float *const *d_scores, int *const *d_kp_idx,
                             const int *kp_caps, int *const *d_kp_counts, void *stream) {
    if (!handles || count <= 0 || !d_kp_idx || !kp_caps || !d_kp_counts) return KPL_ERR_INVALID_ARG;
    for (int k = 0; k < count; ++k)
        if (!handles[k]) return KPL_ERR_INVALID_ARG;
    kpl_detector *h0 = handles[0];
    if (count > kMaxBatch) return fail(h0, KPL_ERR_INVALID_ARG, "at most %d views per batch", kMaxBatch);
    int rc = use_device(h0);
    if (rc) return rc;
    for (int k = 0; k < count; ++k) {
        kpl_detector *h = handles[k];
        if (h->device != h0->device) return fail(h0, KPL_ERR_INVALID_ARG, "all views of a batch must live on one device");
        for (int j = 0; j < k; ++j)
            if (handles[j] == h) return fail(h0, KPL_ERR_INVALID_ARG, "a handle appears twice in the batch");
    }
    return run_batch(handles, count, d_scores, d_kp_idx, kp_caps, d_kp_counts, nullptr, true, (hipStream_t)stream);
}

int kpl_compute_batch_keypoints_device(kpl_detector *const *handles, int count, int *const *d_kp_idx,
                                       float *const *d_kp_scores, const int *kp_caps, int *const *d_kp_counts, void *stream) {
    if (!handles || count <= 0 || !d_kp_idx || !kp_caps || !d_kp_counts) return KPL_ERR_INVALID_ARG;
    for (int k = 0; k < count; ++k)
        if (!handles[k]) return KPL_ERR_INVALID_ARG;
    kpl_detector *h0 = handles[0];
    if (count > kMaxBatch) return fail(h0, KPL_ERR_INVALID_ARG, "at most %d views per batch", kMaxBatch);
    int rc = use_device(h0);
    if (rc) return rc;
    float *scratch[kMaxBatch];               // the response of every point stays in a scratch array of the handle
    for (int k = 0; k < count; ++k) {
        kpl_detector *h = handles[k];
        if (h->device != h0->device) return fail(h0, KPL_ERR_INVALID_ARG, "all views of a batch must live on one device");
        for (int j = 0; j < k; ++j)
            if (handles[j] == h) return fail(h0, KPL_ERR_INVALID_ARG, "a handle appears twice in the batch");
        // (grow-only; an earlier call may still write the old array: released in the order of the stream)
        KPL_HIP(h0, h->out_scores.ensure(sizeof(float) * (size_t)(h->n > 0 ? h->n : 1), (hipStream_t)stream, h->parked));
        scratch[k] = h->out_scores.as<float>();
    }
    return run_batch(handles, count, scratch, d_kp_idx, kp_caps, d_kp_counts, nullptr, true, (hipStream_t)stream, d_kp_scores);
}

// computePointsForTrainingFeatures for up to kMaxBatch bound views in one go: the indices of the views that need one are
// built in ONE batch of index launches, the features of all views in one launch (blockIdx.y = view)
static int features_batch(kpl_detector *const *handles, int count, const int *const *d_indices, const int *m,
                          float *const *d_features, hipStream_t st) {
    kpl_detector *h0 = handles[0];
    Batch idx{}, fix{};
    QueryBatch qb{};
    qb.nviews = count;
    bool rebuilt[kMaxBatch] = {};
    for (int k = 0; k < count; ++k) {
        kpl_detector *h = handles[k];
        if (m[k] < 0 || (m[k] > 0 && (!d_indices[k] || !d_features[k]))) return fail(h0, KPL_ERR_INVALID_ARG, "view %d: null index or feature buffer", k);
        int rc = check_params_for_compute(h, false);
        if (!rc && !h->bound) rc = fail(h, KPL_ERR_NO_CLOUD, "no cloud bound");
        ViewDev v{};
        if (!rc) rc = prepare_index(h, false, v, st);
        if (rc) {
            if (h != h0) fail(h0, rc, "view %d: %s", k, kpl_last_error(h));
            return rc;
        }
        v.want_pos_of = 1;
        if (!index_is_current(h)) {
            idx.view[idx.nviews++] = v;
            rebuilt[k] = true;
        } else if (!h->pos_of_valid) {
            fix.view[fix.nviews++] = v;
        }
        QueryView &q = qb.view[k];
        q.pts = v.pts;
        q.nrm = v.nrm;
        q.nrmsrc = h->d_nrm;
        q.ns = (unsigned)h->ns;
        q.cell_start = v.cell_start;
        q.pos_of = v.pos_of;
        q.ds = v.ds;
        q.f = make_feat(h->prm);
        q.query = d_indices[k];
        q.m = m[k];
        q.n = h->n;
        q.out = d_features[k];
    }
    if (idx.nviews) launch_index(idx, st);
    if (fix.nviews) launch_pos_of(fix, st);
    launch_features(qb, st);
    KPL_HIP(h0, hipGetLastError());
    for (int k = 0; k < count; ++k) {
        if (rebuilt[k]) index_was_built(handles[k], false);
        else handles[k]->pos_of_valid = true;
    }
    return KPL_OK;
}

int kpl_compute_features_device(kpl_detector *h, const int *d_indices, int m, float *d_features, void *stream) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (m < 0 || (m > 0 && (!d_indices || !d_features))) return fail(h, KPL_ERR_INVALID_ARG, "null index or feature buffer");
    int rc = use_device(h);
    if (rc) return rc;
    return features_batch(&h, 1, &d_indices, &m, &d_features, (hipStream_t)stream);
}

int kpl_compute_features_batch_device(kpl_detector *const *handles, int count, const int *const *d_indices, const int *m,
                                      float *const *d_features, void *stream) {
    if (!handles || count <= 0 || !d_indices || !m || !d_features) return KPL_ERR_INVALID_ARG;
    for (int k = 0; k < count; ++k)
        if (!handles[k]) return KPL_ERR_INVALID_ARG;
    kpl_detector *h0 = handles[0];
    if (count > kMaxBatch) return fail(h0, KPL_ERR_INVALID_ARG, "at most %d views per batch", kMaxBatch);
    int rc = use_device(h0);
    if (rc) return rc;
    for (int k = 0; k < count; ++k) {
        if (handles[k]->device != h0->device) return fail(h0, KPL_ERR_INVALID_ARG, "all views of a batch must live on one device");
        for (int j = 0; j < k; ++j)
            if (handles[j] == handles[k]) return fail(h0, KPL_ERR_INVALID_ARG, "a handle appears twice in the batch");
    }
    return features_batch(handles, count, d_indices, m, d_features, (hipStream_t)stream);
}

int kpl_detect(kpl_detector *h, const void *xyz, size_t xyz_stride, const void *normals, size_t normals_stride,
               int n, float *scores_out, int *kp_idx_out, int kp_cap, int *kp_count) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (!kp_count || kp_cap < 0 || (kp_cap > 0 && !kp_idx_out)) return fail(h, KPL_ERR_INVALID_ARG, "bad keypoint output buffers");
    *kp_count = 0;
    int rc = check_params_for_compute(h, true);
    if (rc) return rc;
    rc = upload_view(h, xyz, xyz_stride, normals, normals_stride, n, true);
    if (rc) return rc;
    return detect_staged(h, n, scores_out, kp_idx_out, nullptr, kp_cap, kp_count);
}

int kpl_detect_keypoints(kpl_detector *h, const void *xyz, size_t xyz_stride, const void *normals,
                         size_t normals_stride, int n, int *kp_idx_out, float *kp_scores_out, int kp_cap,
                         int *kp_count) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (!kp_count || kp_cap < 0 || (kp_cap > 0 && !kp_idx_out)) return fail(h, KPL_ERR_INVALID_ARG, "bad keypoint output buffers");
    *kp_count = 0;
    int rc = check_params_for_compute(h, true);
    if (rc) return rc;
    rc = upload_view(h, xyz, xyz_stride, normals, normals_stride, n, true);
    if (rc) return rc;
    return detect_staged(h, n, nullptr, kp_idx_out, kp_scores_out, kp_cap, kp_count);
}

int kpl_host_staging(kpl_detector *h, int n, size_t xyz_stride, size_t normals_stride, void **xyz, void **normals) {
    if (!h || !xyz || !normals) return KPL_ERR_INVALID_ARG;
    *xyz = *normals = nullptr;
    if (n < 0) return fail(h, KPL_ERR_INVALID_ARG, "negative point count");
    if (xyz_stride < 12 || normals_stride < 12 || (xyz_stride & 3) || (normals_stride & 3))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be multiples of 4 and >= 12 bytes");
    int rc = use_device(h);
    if (!rc) rc = streams_ready(h);
    if (rc) return rc;
    const size_t nn = (size_t)(n > 0 ? n : 1), bx = nn * xyz_stride, bn = nn * normals_stride;
    if (bx > h->hs_xyz_cap || bn > h->hs_nrm_cap) KPL_HIP(h, hipStreamSynchronize(h->copy_stream));
    if (bx > h->hs_xyz_cap) {
        if (h->hs_xyz) (void)hipHostFree(h->hs_xyz);
        h->hs_xyz = nullptr;
        h->hs_xyz_cap = 0;
        KPL_HIP(h, hipHostMalloc(&h->hs_xyz, bx + bx / 4, hipHostMallocDefault));
        h->hs_xyz_cap = bx + bx / 4;
    }
    if (bn > h->hs_nrm_cap) {
        if (h->hs_nrm) (void)hipHostFree(h->hs_nrm);
        h->hs_nrm = nullptr;
        h->hs_nrm_cap = 0;
        KPL_HIP(h, hipHostMalloc(&h->hs_nrm, bn + bn / 4, hipHostMallocDefault));
        h->hs_nrm_cap = bn + bn / 4;
    }
    h->hs_xs = xyz_stride;
    h->hs_ns = normals_stride;
    h->hs_n = n;
    *xyz = h->hs_xyz;
    *normals = h->hs_nrm;
    return KPL_OK;
}

int kpl_detect_keypoints_staged(kpl_detector *h, int *kp_idx_out, float *kp_scores_out, int kp_cap, int *kp_count) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (!kp_count || kp_cap < 0 || (kp_cap > 0 && !kp_idx_out)) return fail(h, KPL_ERR_INVALID_ARG, "bad keypoint output buffers");
    *kp_count = 0;
    if (h->hs_n < 0) return fail(h, KPL_ERR_NO_CLOUD, "kpl_host_staging has not been called");
    int rc = check_params_for_compute(h, true);
    if (rc) return rc;
    rc = use_device(h);
    if (!rc) rc = streams_ready(h);
    if (rc) return rc;
    const int n = h->hs_n;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    KPL_HIP(h, h->stage_xyz.ensure(nn * h->hs_xs));
    KPL_HIP(h, h->stage_nrm.ensure(nn * h->hs_ns));
    if (n > 0) {
        // pinned memory: true DMA, asynchronous.  Points first, then the normals, both on the copy stream; the
        // compute stream starts on the points as soon as they have landed and meets the normals at the scatter
        KPL_HIP(h, hipMemcpyAsync(h->stage_xyz.p, h->hs_xyz, (size_t)(n - 1) * h->hs_xs + 12, hipMemcpyHostToDevice, h->copy_stream));
        KPL_HIP(h, hipEventRecord(h->ev_xyz, h->copy_stream));
        KPL_HIP(h, hipMemcpyAsync(h->stage_nrm.p, h->hs_nrm, (size_t)(n - 1) * h->hs_ns + 12, hipMemcpyHostToDevice, h->copy_stream));
        KPL_HIP(h, hipEventRecord(h->ev_nrm, h->copy_stream));
        KPL_HIP(h, hipStreamWaitEvent(h->stream, h->ev_xyz, 0));
    }
    h->d_xyz = h->stage_xyz.as<char>();
    h->d_nrm = h->stage_nrm.as<char>();
    h->xs = h->hs_xs;
    h->ns = h->hs_ns;
    h->n = n;
    h->bound = true;
    h->index_valid = false;
    if (n > 0) estimate_neighborhood(h, h->hs_xyz, h->hs_xs, n);      // a first call only; the copies are on their way meanwhile
    return detect_staged(h, n, nullptr, kp_idx_out, kp_scores_out, kp_cap, kp_count, n > 0 ? h->ev_nrm : nullptr);
}

int kpl_reserve(kpl_detector *h, int n_points, size_t xyz_stride, size_t normals_stride) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (n_points < 0 || n_points >= (1 << 28)) return fail(h, KPL_ERR_INVALID_ARG, "n_points must be in 0 .. 2^28 - 1");
    if ((xyz_stride && (xyz_stride < 12 || (xyz_stride & 3))) || (normals_stride && (normals_stride < 12 || (normals_stride & 3))))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be 0 (no staging arrays) or multiples of 4 and >= 12 bytes");
    int rc = use_device(h);
    if (!rc) rc = streams_ready(h);
    if (rc) return rc;
    const size_t nn = (size_t)(n_points > 0 ? n_points : 1);
    hipStream_t st = h->stream;
    rc = enter_stream(h, st);
    if (rc) return rc;
    // the staging arrays of the host entry points belong to the bound view when it is a host view: never replaced here
    const bool staged_view = h->bound && h->d_xyz == h->stage_xyz.as<char>();
    if (!staged_view) {
        if (xyz_stride) KPL_HIP(h, h->stage_xyz.ensure(nn * xyz_stride));
        if (normals_stride) KPL_HIP(h, h->stage_nrm.ensure(nn * normals_stride));
    }
    rc = ensure_index_tables(h, n_points, st);
    if (!rc && h->prm.n_annulus >= 1 && h->prm.n_bins >= 1 && (int64_t)h->prm.n_annulus * h->prm.n_bins <= 255)
        rc = ensure_detect_scratch(h, n_points, st);
    if (rc) return rc;
    KPL_HIP(h, h->out_kp.ensure(sizeof(int) * nn));
    KPL_HIP(h, h->out_kp_score.ensure(sizeof(float) * nn));
    KPL_HIP(h, h->out_count.ensure(sizeof(int)));
    KPL_HIP(h, hipStreamSynchronize(st));       // a set-up call: what it cleared is clear when it returns
    return KPL_OK;
}

int kpl_sync_status(kpl_detector *h, void *stream) {
    if (!h) return KPL_ERR_INVALID_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    return sync_status(h, (hipStream_t)stream);
}

int kpl_compute_features(kpl_detector *h, const void *xyz, size_t xyz_stride, const void *normals,
                         size_t normals_stride, int n, const int *indices, int m, float *features_out) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (m < 0 || (m > 0 && (!indices || !features_out))) return fail(h, KPL_ERR_INVALID_ARG, "null index or feature buffer");
    int rc = check_params_for_compute(h, false);
    if (rc) return rc;
    rc = upload_view(h, xyz, xyz_stride, normals, normals_stride, n);
    if (rc) return rc;
    if (m == 0) return KPL_OK;
    const size_t F = (size_t)h->prm.n_annulus * h->prm.n_bins;
    KPL_HIP(h, h->stage_idx.ensure(sizeof(int) * (size_t)m));
    KPL_HIP(h, h->stage_feat.ensure(sizeof(float) * (size_t)m * F));
    KPL_HIP(h, hipMemcpyAsync(h->stage_idx.p, indices, sizeof(int) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    for (int attempt = 0;; ++attempt) {
        rc = kpl_compute_features_device(h, h->stage_idx.as<int>(), m, h->stage_feat.as<float>(), h->stream);
        if (rc) return rc;
        rc = sync_status(h, h->stream);
        if (rc == KPL_ERR_RETRY && attempt == 0) continue;
        if (rc) return rc;
        break;
    }
    KPL_HIP(h, hipMemcpy(features_out, h->stage_feat.p, sizeof(float) * (size_t)m * F, hipMemcpyDeviceToHost));
    return KPL_OK;
}

int kpl_enable_timing(kpl_detector *h, int enable) {
    if (!h) return KPL_ERR_INVALID_ARG;
    h->timing = enable != 0;
    h->spans.clear();
    h->ev_used = 0;
    return KPL_OK;
}

int kpl_get_timing(kpl_detector *h, kpl_timing *out) {
    if (!h || !out) return KPL_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    int rc = use_device(h);
    if (rc) return rc;
    for (const auto &sp : h->spans) {
        KPL_HIP(h, hipEventSynchronize(h->ev_pool[sp.b]));
        float ms = 0.0f;
        KPL_HIP(h, hipEventElapsedTime(&ms, h->ev_pool[sp.a], h->ev_pool[sp.b]));
        if (sp.phase == 0) out->index_ms += ms;
        else if (sp.phase == 1) { out->score_ms += ms; out->feature_ms += ms; out->calls++; }
        else if (sp.phase == 3) { out->score_ms += ms; out->forest_ms += ms; }
        else out->nms_ms += ms;
    }
    h->spans.clear();
    h->ev_used = 0;
    out->walk = h->last_walk;
    out->lanes_per_point = h->last_lanes;
    out->accept_words = h->last_words;
    return KPL_OK;
}

int kpl_collect_stats(kpl_detector *h, kpl_stats *out, void *stream) {
    if (!h || !out) return KPL_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    int rc = use_device(h);
    if (!rc) rc = streams_ready(h);      // (out_kp below is also the set-up thread's: a handle used through the device entry points
                                         //  only has not joined it yet)
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    KPL_HIP(h, h->stats.ensure(sizeof(StatsDev)));
    KPL_HIP(h, hipMemsetAsync(h->stats.p, 0, sizeof(StatsDev), st));
    const size_t nn = (size_t)(h->n > 0 ? h->n : 1);
    KPL_HIP(h, h->out_kp.ensure(sizeof(int) * nn));
    KPL_HIP(h, h->out_count.ensure(sizeof(int)));
    rc = detect_on_device(h, nullptr, h->out_kp.as<int>(), h->n, h->out_count.as<int>(), st, h->stats.as<StatsDev>());
    if (rc) return rc;
    StatsDev sd;
    KPL_HIP(h, hipMemcpyAsync(h->h_count, h->out_count.p, sizeof(int), hipMemcpyDeviceToHost, st));
    rc = sync_status(h, st);
    if (rc) return rc;
    KPL_HIP(h, hipMemcpy(&sd, h->stats.p, sizeof(sd), hipMemcpyDeviceToHost));
    out->n_points = h->n;
    out->n_scored = (int64_t)sd.n_scored;
    out->n_thresholded = (int64_t)sd.n_thresholded;
    out->sum_kf = (int64_t)sd.sum_kf;
    out->sum_kn = (int64_t)sd.sum_kn;
    out->sum_depth = (int64_t)sd.sum_depth;
    out->n_keypoints = h->h_count[0];
    out->n_cells = h->h_state->grid.ncells;
    return KPL_OK;
}

int kpl_estimate_normals_device(kpl_detector *h, int k_search, double radius_search, const float *viewpoint,
                                void *d_normals, size_t normals_stride, void *d_curvature, size_t curvature_stride,
                                void *stream) {
    if (!h) return KPL_ERR_INVALID_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    return normals_on_device(h, k_search, radius_search, viewpoint, d_normals, normals_stride, d_curvature,
                             curvature_stride, (hipStream_t)stream);
}

int kpl_estimate_normals(kpl_detector *h, const void *xyz, size_t xyz_stride, int n, int k_search, double radius_search,
                         const float *viewpoint, void *normals_out, size_t normals_stride, void *curvature_out,
                         size_t curvature_stride) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (n > 0 && !normals_out) return fail(h, KPL_ERR_INVALID_ARG, "null normals buffer");
    if (normals_stride < 12 || (normals_stride & 3) || (curvature_out && (curvature_stride < 4 || (curvature_stride & 3))))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be multiples of 4 (normals >= 12 bytes)");
    // staged without normals: the index build copies whatever sits at the normal pointer
    int rc = upload_view(h, xyz, xyz_stride, xyz, xyz_stride, n);
    if (rc) return rc;
    UnbindOnExit unbind{h};
    h->d_nrm = h->d_xyz;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    KPL_HIP(h, h->stage_feat.ensure(sizeof(float) * 4 * nn));       // (nx, ny, nz, curvature) per point
    hipStream_t st = h->stream;          // where upload_view put the copies
    for (int attempt = 0;; ++attempt) {
        rc = normals_on_device(h, k_search, radius_search, viewpoint, h->stage_feat.p, 16,
                               (char *)h->stage_feat.p + 12, 16, st);
        if (rc) return rc;
        rc = sync_status(h, st);
        if (rc == KPL_ERR_RETRY && attempt == 0) continue;
        if (rc) return rc;
        break;
    }
    std::vector<float> tmp(4 * nn);
    KPL_HIP(h, hipMemcpy(tmp.data(), h->stage_feat.p, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {
        memcpy((char *)normals_out + (size_t)i * normals_stride, &tmp[4 * (size_t)i], 12);
        if (curvature_out) memcpy((char *)curvature_out + (size_t)i * curvature_stride, &tmp[4 * (size_t)i + 3], 4);
    }
    return KPL_OK;
}

// pcl::IntegralImageNormalEstimation (SIMPLE_3D_GRADIENT) on an organized cloud in device memory
static int organized_normals_on_device(kpl_detector *h, const void *d_xyz, size_t xs, int width, int height,
                                       float smoothing, const float *viewpoint, void *d_normals, size_t ns, void *d_curv,
                                       size_t cs, hipStream_t st) {
    if (width < 0 || height < 0 || (long long)width * (long long)height > (1ll << 28))
        return fail(h, KPL_ERR_INVALID_ARG, "width x height must be in 0 .. 2^28");
    if ((long long)width * height > 0 && (!d_xyz || !d_normals)) return fail(h, KPL_ERR_INVALID_ARG, "null cloud or normals buffer");
    if (xs < 12 || ns < 12 || (xs & 3) || (ns & 3) || (d_curv && (cs < 4 || (cs & 3))))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be multiples of 4 (points and normals >= 12 bytes)");
    if (!(smoothing >= 0.0f) || !(smoothing < 1.0e6f)) return fail(h, KPL_ERR_INVALID_ARG, "normal smoothing size must be in [0, 1e6)");
    if (width == 0 || height == 0) return KPL_OK;
    KPL_HIP(h, h->org_scratch.ensure(organized_normals_scratch_bytes(width, height)));
    OrganizedView v{};
    v.xyz = (const char *)d_xyz;
    v.xs = xs;
    v.W = width;
    v.H = height;
    v.smoothing = smoothing;
    for (int k = 0; k < 3; ++k) v.vp[k] = viewpoint ? viewpoint[k] : 0.0f;
    v.normals = (char *)d_normals;
    v.ns = ns;
    v.curvature = (char *)d_curv;
    v.cs = cs;
    launch_organized_normals(v, h->org_scratch.p, st);
    KPL_HIP(h, hipGetLastError());
    return KPL_OK;
}

int kpl_estimate_normals_organized_device(kpl_detector *h, const void *d_xyz, size_t xyz_stride, int width, int height,
                                          float normal_smoothing_size, const float *viewpoint, void *d_normals,
                                          size_t normals_stride, void *d_curvature, size_t curvature_stride, void *stream) {
    if (!h) return KPL_ERR_INVALID_ARG;
    int rc = use_device(h);
    if (rc) return rc;
    return organized_normals_on_device(h, d_xyz, xyz_stride, width, height, normal_smoothing_size, viewpoint, d_normals,
                                       normals_stride, d_curvature, curvature_stride, (hipStream_t)stream);
}

int kpl_estimate_normals_organized(kpl_detector *h, const void *xyz, size_t xyz_stride, int width, int height,
                                   float normal_smoothing_size, const float *viewpoint, void *normals_out,
                                   size_t normals_stride, void *curvature_out, size_t curvature_stride) {
    if (!h) return KPL_ERR_INVALID_ARG;
    if (width < 0 || height < 0 || (long long)width * (long long)height > (1ll << 28))
        return fail(h, KPL_ERR_INVALID_ARG, "width x height must be in 0 .. 2^28");
    const int n = width * height;
    if (n > 0 && (!xyz || !normals_out)) return fail(h, KPL_ERR_INVALID_ARG, "null cloud or normals buffer");
    if (xyz_stride < 12 || (xyz_stride & 3) || normals_stride < 12 || (normals_stride & 3) ||
        (curvature_out && (curvature_stride < 4 || (curvature_stride & 3))))
        return fail(h, KPL_ERR_INVALID_ARG, "strides must be multiples of 4 (points and normals >= 12 bytes)");
    int rc = use_device(h);
    if (!rc) rc = streams_ready(h);
    if (rc) return rc;
    if (n == 0) return KPL_OK;
    const size_t nn = (size_t)n;
    const bool staged_view = h->bound && h->d_xyz == h->stage_xyz.as<char>();   // a bound HOST view lives in stage_xyz
    KPL_HIP(h, h->stage_xyz.ensure(nn * xyz_stride));
    KPL_HIP(h, h->stage_feat.ensure(sizeof(float) * 4 * nn));       // (nx, ny, nz, curvature) per pixel
    hipStream_t st = h->stream;
    KPL_HIP(h, hipMemcpyAsync(h->stage_xyz.p, xyz, (nn - 1) * xyz_stride + 12, hipMemcpyHostToDevice, st));
    if (staged_view) {                  // its staging buffer was just overwritten (a view bound with
        h->bound = false;               // kpl_bind_cloud_device is the caller's memory: untouched)
        h->index_valid = false;
    }
    rc = organized_normals_on_device(h, h->stage_xyz.p, xyz_stride, width, height, normal_smoothing_size, viewpoint,
                                     h->stage_feat.p, 16, (char *)h->stage_feat.p + 12, 16, st);
    if (rc) return rc;
    std::vector<float> tmp(4 * nn);
    KPL_HIP(h, hipMemcpyAsync(tmp.data(), h->stage_feat.p, sizeof(float) * 4 * nn, hipMemcpyDeviceToHost, st));
    KPL_HIP(h, hipStreamSynchronize(st));
    for (size_t i = 0; i < nn; ++i) {
        memcpy((char *)normals_out + i * normals_stride, &tmp[4 * i], 12);
        if (curvature_out) memcpy((char *)curvature_out + i * curvature_stride, &tmp[4 * i + 3], 4);
    }
    return KPL_OK;
}

int kpl_cloud_resolution(kpl_detector *h, const void *xyz, size_t xyz_stride, int n, double *resolution) {
    if (!h || !resolution) return KPL_ERR_INVALID_ARG;
    *resolution = 0.0;
    // the view is bound without normals: the index build copies whatever sits at the normal
    // pointer, which this entry point never looks at
    int rc = upload_view(h, xyz, xyz_stride, xyz, xyz_stride, n);
    if (rc) return rc;
    UnbindOnExit unbind{h};
    h->d_nrm = h->d_xyz;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    KPL_HIP(h, h->stage_feat.ensure(sizeof(float) * nn));
    KPL_HIP(h, h->out_scores.ensure(3 * sizeof(double)));
    KPL_HIP(h, h->out_kp.ensure(resolution_scratch_bytes()));
    hipStream_t st = h->stream;          // where upload_view put the copies
    for (int attempt = 0;; ++attempt) {
        rc = build_index(h, st, true);
        if (rc) return rc;
        launch_resolution(h->pts.as<float4>(), h->cell_start.as<int>(), h->pos_of.as<int>(), h->dstate.as<DevState>(), n,
                          h->stage_feat.as<float>(), h->out_scores.as<double>(), h->out_kp.p, st);
        KPL_HIP(h, hipGetLastError());
        rc = sync_status(h, st);
        if (rc == KPL_ERR_RETRY && attempt == 0) continue;
        if (rc) return rc;
        break;
    }
    double out[2];
    KPL_HIP(h, hipMemcpy(out, h->out_scores.p, sizeof(out), hipMemcpyDeviceToHost));
    if (out[1] > 0.0) *resolution = out[0] / out[1];                          // hpp:145-148
    return KPL_OK;
}

}  // extern "C"
