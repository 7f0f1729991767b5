// kernels.h -- launch wrappers of the gfx950 kernels of libkpl (definitions in kernels.hip).
//
// Data layout in HBM (per bound view, all owned by the handle):
//   pts[s]   float4  xyz of the s-th finite point in canonical storage order, w = bits of its
//                    original index                       (16 B, one dwordx4 load per candidate)
//   nrm[s]   float4  its normal, w = 1.0f if the normal is finite else 0.0f
//   cell_start[c]    first storage position of grid cell c, c = (cz*ny + cy)*nx + cx; a run of
//                    cells along x is therefore ONE contiguous range of pts/nrm
//   pos_of[i]        storage position of original point i, -1 if its xyz is not finite (built on demand)
//   score_sorted[s]  forest response in storage order (what the NMS kernel gathers)
//   flags[i]         1 if original point i is a keypoint (compacted in ascending i)
// Canonical storage order = ascending (cell id, original index); it is what makes the float
// accumulation order of the histogram identical to the oracle's.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace kpl {

struct GridDesc {
    float mn[3];
    float h;
    int dims[3];
    int ncells;
};

struct FeatDesc {
    int A, B, F;
    float A1f, B1f;  // (float)(A - 1), (float)(B - 1): the clamp of the soft-assignment index in the float domain
    float support;   // (float)radius_search, the `support` argument of findAnnulusPair
    float ann_dim;   // support / A
    float ann_half;  // ann_dim / 2
    float ann_rdim;  // RN(1 / ann_dim), for the exact 3-instruction division (kernels.hip div_rn)
    float bin_dim;   // 2 / (float)B
    float bin_half;
    float bin_rdim;
    float r2;        // (float)(r*r), product in double -- KdTreeFLANN::radiusSearch
    float rr;        // (float)(r*(1+2^-10)): half width of the cell box that is searched
    int sorted;      // neighbor order of the feature loop: 0 = canonical (cell id, index), 1 = ascending (d2, index)
    // how the canonical order is walked -- never WHAT is computed: every combination gives the same bits (kernels.hip)
    int walk;        // 0: search and drain alternate, accept words in LDS (point_features); 1: two passes, the accept words of the whole walk through global memory (large neighborhoods).
                     // sorted order with walk = 1 ("sorted words", ~100 .. ~420 neighbors per point): the same search pass (no neighbor dropped), then
                     // sorted_words_kernel -- eight lanes per point expand the point's word list into positions, sort them through 32-bit
                     // stand-ins in registers and add in order
    int lanes;       // lanes per point: 2 or 4 (sorted words: 8)
    int words;       // one-kernel walk: accept words a point collects between two drains (0 = 24; fewer for small neighborhoods, kernels.hip accept_words)
    int lcap;        // sorted-search mode: positions per point of feature_sorted_kernel's lists in LDS (<= 128; 0 = 128); sorted words: 256 or 512
    int all_large;   // sorted-search mode: every point goes to the collect / add kernels without trying the register sort first
                     // (the handle's last call listed nearly all of them anyway -- after searching for most)
};

struct NmsDesc {
    float r2, rr;
    double thr;
    int non_maxima;
    int draws_remove;     // non_maxima_draws_remove_
    float draws_thr;      // non_maxima_draws_threshold_
    int scan_poll_limit;  // look-back polls of the compaction's single-pass scan before the call is failed (2^22; < 0: every block
                          // but the first gives up at once -- the failure path, forced by tests through kpl_debug_set_scan_poll_limit)
};

// candidates of the NMS stage: storage positions of the points whose score passed the threshold,
// appended by the score kernel in arbitrary order (the NMS predicate is order independent)
struct NmsList {
    int *list;    // [n]
    int *count;   // zero between calls (compact_kernel resets it)
};

struct ForestDev {
    const uint2 *nodes;      // level-major (forest.h): node t is the root of tree t, the first k nodes are the top of every tree
    int ntrees;
    int nnodes;              // slots
    int ntop;                // slots of the level-major top part; slots >= ntop are 8-slot blocks (forest.h)
    int order_free;          // FlatForest::order_free: the trees of a point may be summed in any order
    int chain;               // FlatForest::chain: the leaf records chain tree t to tree t + chain
};

struct StatsDev {
    unsigned long long sum_kf, sum_kn, sum_depth, n_scored, n_thresholded;
};

struct DevState;

// Everything the kernels need to know about one view.  Every kernel of the pipeline takes a Batch
// of these and picks its view with blockIdx.y, so a call over k views costs the same ~15 launches
// as a call over one (the single-view entry points are batches of one).
struct ViewDev {
    // input as bound by the caller (byte strides)
    const char *xyz, *nrmsrc;
    unsigned xs, ns;
    int n;
    // index ("initCompute")
    DevState *ds;
    int cells_cap;               // capacity of cell_start
    float cell;                  // cell edge; <= 0: derived from the bounding box (cloud resolution)
    float origin[3];             // grid origin when has_origin, else the minimum of the finite points
    int has_origin;
    int *cid;                    // [n] cell of original point i, -1 if not finite
    int *cell_start, *scan_tmp;
    int *btable, *btotal, *bstart;   // index sort, level 1: (chunk, bin) counts -> offsets; points / first position of each bin
    float4 *rec;                 // [2 n] records (x, y, z, index)(nx, ny, nz, cell), bucket by bucket, index order inside
    float4 *pts, *nrm;           // canonical storage order
    int *pos_of;                 // [n] original index -> storage position (-1: no cell); complete only if want_pos_of
    int want_pos_of;
    // scoring ("runForest")
    FeatDesc f;
    ForestDev forest;
    NmsDesc nd;
    float *feat;                 // scratch, feat_bytes(n, F): the feature rows between the two scoring kernels
    float *score_sorted;         // [n] out, storage order
    float *scores;               // [n] out, original order (may be null)
    int *flags, *prefix;         // [n+1] keypoint flags in original order and their scan
    unsigned long long *scan_state;   // one word per 4096 flags: the single-pass scan of the compaction (kernels.hip)
    NmsList cand;                // points that passed the threshold
    // draws pass + compaction ("detectKeypoints")
    int *draw_list, *draw_count, *skip;    // draw_list: [n] listed maxima, then [n] adjacency counts, then [n x kDrawAdj] adjacency (draws pass)
    int *kp_idx;
    float *kp_score;             // [kp_cap] forest response of each keypoint (may be null; needs `scores`)
    int kp_cap;
    int *kp_count;
    StatsDev *stats;             // null unless counters are collected
    // sorted-search mode, large neighborhoods (kernels.hip "sorted mode, large neighborhoods"): scoreable points whose search
    // box holds more than kLargeCand candidates, their neighbor keys sorted by (d2, index) in one array, a segment per point
    int *large_list;                 // [2 n] storage positions of those points (any order); their number: DevState::large_count.
                                     // From [n] on: the ones with more keys than a wave sorts (DevState::huge_count)
    unsigned long long *sort_keys;   // [key_cap] the segments
    unsigned *seg_start;             // [n] by storage position: first key of the point's segment
    int *seg_len;                    // [n] its length (0: none, or the keys did not fit -> DevState::status)
    unsigned long long key_cap;
    // sorted order through the word lists (FeatDesc::sorted && walk == 1): a view then needs BOTH arrays -- the accept words of
    // every point here, the key segments of the points that overflow 256 keys in sort_keys -- and the words' own per-point tables
    uint2 *words;                    // [word_cap] the blocks of the search pass
    unsigned long long word_cap;
    unsigned *wseg_start;            // [n] first entry of the point's word list
    int *wseg_len;                   // [n] its entries
};

constexpr int kMaxBatch = 8;     // views per batched launch (bounded by the 4 KB kernel argument block)
struct Batch {
    int nviews;
    ViewDev view[kMaxBatch];
};
static_assert(sizeof(Batch) <= 4096, "a Batch travels as kernel arguments: 4 KB is the limit");

// Device-resident state of one handle: the grid descriptor is computed ON the device from the
// bounding box, so the host never waits between the kernels of a call.
constexpr int kStatusOk = 0, kStatusGridTooLarge = 1, kStatusCellCapacity = 2, kStatusBadOrigin = 3, kStatusKeyCapacity = 4;
constexpr long long kMaxGridCells = 1ll << 28;
constexpr int kBuckets = 1024;   // buckets of the index sort (kernels.hip "Index build")
constexpr int kDrawRounds = 8;       // parallel rounds of the draws pass over the adjacency rows, before the pipelined rest (kernels.hip)
constexpr int kDrawAdj = 32;         // lower-index neighbors an entry of the draws pass keeps for the sequential rest
struct DevState {
    GridDesc grid;        // written by grid_setup_kernel, read by every later kernel
    int status;           // kStatus*: on failure the grid is empty and kp_count becomes -1
    int ncells_needed;    // what this view needs (to grow the cell tables before a retry)
    int bshift, nbuckets; // index sort: nbuckets buckets of 2^bshift consecutive cells
    uint32_t bbox[6];     // order-preserving encoded min / max accumulators (self re-arming)
    uint32_t scan_epoch;  // tag of the next detect call in the words of scan_state (advanced on the device, never 0)
    int scan_fail;        // set by a block of compact_scan_kernel whose look-back gave up: the call failed (kpl_sync_status -> KPL_ERR_INTERNAL, which clears it)
    int draws_left[kDrawRounds + 2];   // draws pass: [0] = listed maxima, [1] = undecided after the adjacency pass, [r + 2] = after round r
    // sorted-search mode, large neighborhoods: both counters are zero between calls (the compaction's last block re-arms them
    // after copying the cursor to keys_needed, which is what kpl_sync_status grows ViewDev::sort_keys to on kStatusKeyCapacity)
    int large_count;                   // points in ViewDev::large_list
    int huge_count;                    // of them, points left to the workgroup kernel (second half of large_list)
    unsigned long long key_cursor;     // keys handed out of ViewDev::sort_keys so far (counts on past key_cap)
    unsigned long long keys_needed;    // key_cursor of the last call
    unsigned long long words_needed;   // ... and the accept-word entries it asked for (32 x the fullest cursor), already part of keys_needed
    // neighborhood size of the view, for the handle's NEXT call: sum of K_f and number of points over a sample of the waves of
    // the feature kernels (one in 64); cumulative, the host takes differences (api.cpp).  On a cache line of their own:
    // every wave of every kernel reads the grid descriptor at the top of this struct, and atomics on its line queue those
    // reads behind them (the bench lost 3 % with the counters next to it)
    alignas(128) unsigned long long kf_sum;
    unsigned long long kf_points;
    // sorted-search mode: the longest neighborhood the register-sort kernel scored since the host last read it (129: a point
    // whose list ran full and was deferred) -- the list capacity of the handle's next launch (api.cpp)
    int kf_max;
    int large_seen;    // sorted-search mode: DevState::large_count of the last call (copied before it is re-armed)
    // two-pass walk (kernels.hip): entries handed out of each 32nd of the word list (ViewDev::sort_keys); zero between calls
    alignas(128) unsigned long long word_cursor[32];
};
void init_dev_state(DevState *host_copy);
// clears bytes (a multiple of 4) at p with a kernel on `st`
void launch_zero(void *p, size_t bytes, hipStream_t st);
// loads the code objects of both kernel files now (kpl_create) instead of under the first launch of a compute()
void preload_code();

// ---- the three stages of compute(), each over every view of the batch ------------------------
// index build ("initCompute"): needs the input + index fields of ViewDev
void launch_index(const Batch &b, hipStream_t st);
// the same in two halves: the kernels that read only the points, then those that also read the normals (a caller
// that uploads the normals on another stream waits for them between the two)
void launch_index_points(const Batch &b, hipStream_t st);
void launch_index_records(const Batch &b, hipStream_t st);
// pos_of[] of views indexed without it (want_pos_of must be set)
void launch_pos_of(const Batch &b, hipStream_t st);
// scoring ("runForest") in two kernels: features of every point -> feat (F x 64 blocks), then the
// forest: scores[i] (original order, may be null) and score_sorted[s]; NaN where not scoreable;
// appends the points that pass the threshold to `cand` (or, without NMS, flags every scoreable point)
void launch_feature_stage(const Batch &b, hipStream_t st);
void launch_forest_stage(const Batch &b, hipStream_t st);
// NMS, draws pass, ordered compaction ("detectKeypoints").  flags[] / cand.count / skip[] must be
// all zero on entry to a detect call; the compaction leaves them zeroed again
void launch_post(const Batch &b, hipStream_t st);

// bytes of ViewDev::scan_state for a view of n points (zeroed once when allocated, never cleared afterwards)
size_t scan_state_bytes(int n);
// ints of the (chunk, bin) table + bin totals + bin starts of the index sort of a view of n points (the two
// arrays of kBuckets + 1 ints each sit at the end)
size_t btable_ints(int n);
// bytes of the feature scratch of a view: F floats per point, one F x 64 block per wave
size_t feat_bytes(int n, int F);
// bytes of pts[] for a view of n points: a search step of the feature code loads a fixed number of
// consecutive candidates from one address, so the array carries that many elements of tail
size_t pts_bytes(int n);
// features of listed points -> out[m*F], for up to kMaxBatch indexed views per launch (computePointsForTrainingFeatures)
// (nrmsrc / ns: the caller's normals in original point order and their byte stride -- read by the sorted-search mode)
struct QueryView {
    const float4 *pts, *nrm;
    const char *nrmsrc;
    unsigned ns;
    const int *cell_start, *pos_of;
    const DevState *ds;
    FeatDesc f;
    const int *query;     // [m] original point indices
    int m, n;
    float *out;           // [m x F]
};
struct QueryBatch {
    int nviews;
    QueryView view[kMaxBatch];
};
void launch_features(const QueryBatch &qb, hipStream_t st);

// cloud resolution: val[n] scratch, out[0] = ordered double sum of the 2nd-NN distances, out[1] = count
// (out holds 3 doubles; scratch holds resolution_scratch_bytes())
void launch_resolution(const float4 *pts, const int *cell_start, const int *pos_of, const DevState *ds,
                       int n, float *val, double *out, void *scratch, hipStream_t st);
size_t resolution_scratch_bytes();

// normals of every point of the indexed view (pcl::NormalEstimation restated, see kernels.hip):
// k > 0: k-search (k <= 32) on any grid; k <= 0: radius search, r2/rr as in FeatDesc, on the grid
// whose cell edge is that radius.  Output at byte strides, in original point order.
void launch_normals(const float4 *pts, const int *cell_start, const int *pos_of, const DevState *ds, int n,
                    int k, float r2, float rr, const float *viewpoint, char *normals, size_t normals_stride,
                    char *curvature, size_t curvature_stride, hipStream_t st);

}  // namespace kpl
