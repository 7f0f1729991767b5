/*
 * kpl.h -- C-ABI of libkpl: the MI355X (gfx950) engine for the scoring path of
 * pcl::keypoints::KeypointLearningDetector (feature -> random forest -> radius NMS).
 *
 * This is the drop-in boundary.  The reference has no FFI; its boundary is the C++ class
 * declared in /root/reference/include/KeypointLearning.h:55-206.  Each entry point below names
 * the reference member it replaces; include/KeypointLearning.h in this repo is the same class
 * re-implemented on top of these calls, and INTEGRATION.md shows the binding a maintainer of
 * the reference would add.
 *
 * Conventions: plain pointers and sizes only; every function returns a kpl_status (0 = OK) and
 * never throws; kpl_last_error() gives the message of the last failure on that handle.  A
 * handle is not thread safe; distinct handles are independent (one handle per view/stream).
 * All arrays are caller owned; the handle owns its forest and (grow-only) device scratch.
 * Streams: the host-buffer entry points (kpl_detect, kpl_detect_keypoints, kpl_compute_features,
 * kpl_estimate_normals*, kpl_cloud_resolution) run on a private non-blocking stream of the handle and
 * return when their results are in the caller's buffers; the *_device entry points only enqueue on the
 * stream they are given.  Both kinds use the handle's scratch: before mixing them on ONE handle, wait for
 * the stream of the earlier *_device call (kpl_sync_status or hipStreamSynchronize) if you need its RESULTS -- the handle's
 * scratch is ordered for you: a call that arrives on another stream than the handle's previous one makes its stream wait
 * (hipStreamWaitEvent, no host wait) for what the handle queued on the earlier stream, which therefore must still exist.
 * Nothing else is ordered implicitly, not even the null stream (and a clear or copy the CALLER issues on the null stream -- hipMemset, hipMemcpy --
 * is not ordered against a non-blocking stream either, nor complete when it returns: hipDeviceSynchronize() before handing
 * such a buffer to a *_device entry point on another stream).  The same holds for two *_device calls of ONE handle on
 * DIFFERENT streams: the scratch -- and its growth, which is ordered by the stream of the call that grows it -- belongs to
 * the handle, so wait for the earlier stream first (one handle per stream is the intended use).
 * There is NO CPU fallback: without a usable HIP device every compute call fails with
 * KPL_ERR_DEVICE.
 */
#ifndef KPL_H
#define KPL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPL_VERSION 150

typedef enum kpl_status {
    KPL_OK = 0,
    KPL_ERR_INVALID_ARG = 1,   /* null pointer, negative size, bad stride, radius <= 0 ...      */
    KPL_ERR_NO_FOREST = 2,     /* detect before loadForest                                     */
    KPL_ERR_FOREST_PARSE = 3,  /* malformed YAML / unsupported forest                          */
    KPL_ERR_VAR_COUNT = 4,     /* n_annulus * n_bins != forest var_count                       */
    KPL_ERR_GRID_TOO_LARGE = 5,/* bounding box / radius needs more than 2^28 cells             */
    KPL_ERR_CAPACITY = 6,      /* kp_cap too small; kp_count still reports the needed size     */
    KPL_ERR_DEVICE = 7,        /* HIP error or no device                                       */
    KPL_ERR_UNSUPPORTED = 8,   /* k-search mode, surface != input, > 255 features ...          */
    KPL_ERR_IO = 9,            /* file cannot be read                                          */
    KPL_ERR_NO_CLOUD = 10,     /* compute before a cloud was bound                             */
    KPL_ERR_RETRY = 11,        /* kpl_sync_status: the view needed larger cell tables; they have
                                  been grown, enqueue the same call again                      */
    KPL_ERR_INTERNAL = 12      /* a device-side consistency check failed (the look-back of the
                                  compaction's scan timed out): the results of the call are
                                  invalid, *d_kp_count is -1 or unspecified; call again         */
} kpl_status;

typedef struct kpl_detector kpl_detector;

/* Parameters = the state set by the reference's ctor and setters
 * (/root/reference/include/KeypointLearning.h:81-90, impl/KeypointLearning.hpp:60-113) plus
 * pcl::Keypoint::setRadiusSearch (used at /root/reference/src/main_test_detector.cpp:130). */
typedef struct kpl_params {
    int n_annulus;                    /* setNAnnulus, default 5                               */
    int n_bins;                       /* setNBins, default 10                                 */
    double radius_search;             /* setRadiusSearch: feature support radius (required)   */
    double non_max_radius;            /* setNonMaxRadius, default 0                           */
    double prediction_th;             /* setPredictionThreshold, default 0.5                  */
    int non_maxima;                   /* setNonMaxima, default 1                              */
    int non_maxima_draws_remove;      /* setNonMaximaDrawsRemove, default 1                   */
    float non_maxima_draws_threshold; /* setNonMaximaDrawsThreshold (uninitialised in the
                                         reference ctor; 0 here)                              */
    int neighbor_order;               /* pcl::Keypoint::setSearchMethod (inherited,
                                         /root/reference/include/KeypointLearning.h:56): the order in which
                                         the feature loop (impl/KeypointLearning.hpp:334-359) meets the
                                         neighbors; element 0 of it is dropped (:336).  See below.        */
} kpl_params;

/* kpl_params::neighbor_order.  The neighbor SET is the same in both (strict d2 < (float)(r*r)); the order
 * decides which neighbor hpp:336 drops and the order of the float additions of the histogram.
 *   CANONICAL  ascending (grid cell id, point index): this engine's deterministic stand-in for the traversal
 *              order of the default pcl::search::KdTree(false) -> FLANN, which cannot be reproduced without
 *              FLANN.  Default.
 *   SORTED     ascending (squared distance, point index): exactly what a caller gets who passes a
 *              pcl::search::KdTree constructed with sorted = true to setSearchMethod (FLANN sorts its radius
 *              result set by (distance, index)).  Element 0 is then the query itself (or a duplicate of it with a
 *              lower index).  This is the order in which results can be compared bit for bit with a PCL build of
 *              the reference.  Cost against CANONICAL (round 6): 2.1-2.4x up to ~125 neighbors per
 *              point, 2.5-3.4x between ~100 and ~460 (word lists, eight lanes per point), ~5x around 500 (a wave / a
 *              workgroup per point), ~2.3x at the ~2 300 of the reference's default radius.  The order of a point's
 *              neighbors is decided by 32-bit stand-ins of their (distance, index) keys wherever those differ, by the
 *              64-bit keys themselves where distances are equal or almost equal -- the same order either way. */
enum { KPL_NEIGHBORS_CANONICAL = 0, KPL_NEIGHBORS_SORTED = 1 };

/* How the feature kernels WALK the canonical order (never what they compute: every choice gives the same bits --
 * tests/test_gpu_walks.py).  AUTO (default): picked per launch from the mean number of neighbors per point that the
 * handle's own earlier calls measured (read back by kpl_sync_status).  Until then, and whenever the feature radius or
 * the size of the view changes: LANES with two lanes per point for the device entry points; the host entry points -- which
 * have the points in hand, and which a drop-in TestDetector run calls exactly once -- estimate the number from the view's
 * bounding box (pi r^2 x points / the product of its two largest extents: a 2.5D view is a surface) and take TWO_PASS on a
 * first call already when that says 600 or more.
 *   LANES     search and drain alternate in one kernel, the accept words of a point in LDS (neighborhoods of up to a few
 *             hundred points)
 *   TWO_PASS  the accept words of a point's whole walk go through global memory, a second kernel drains every list in one
 *             go (the reference's default radius: ~2 300 neighbors per point).  Needs ~0.3 x 8 bytes per neighbor of
 *             scratch; a first call whose scratch is too small returns KPL_ERR_RETRY like a view whose grid has grown
 * lanes_per_point: 2 or 4. */
enum { KPL_WALK_AUTO = -1, KPL_WALK_LANES = 0, KPL_WALK_TWO_PASS = 1 };
int kpl_set_feature_walk(kpl_detector *h, int walk, int lanes_per_point);
/* what the next launch would use, and the mean neighbors per point it is based on (measured by the handle's last call, or
 * estimated from the bounding box by a first host call; < 0: neither yet) */
int kpl_get_feature_walk(const kpl_detector *h, int *walk, int *lanes_per_point, double *mean_neighbors);

/* What the handle's LAST scoring launch took -- a plain read of the handle, no wait, nothing cleared (kpl_get_timing, which
 * carries the same three walk fields, waits for its events and clears the recorded times). */
typedef struct kpl_launch_info {
    int walk;               /* KPL_WALK_LANES / KPL_WALK_TWO_PASS; -1: sorted order (per-point lists / wave / workgroup kernels), or no
                               call yet; sorted order with KPL_WALK_TWO_PASS: through the word lists, eight lanes per point   */
    int lanes_per_point;
    int accept_words;       /* KPL_WALK_LANES: accept words a point collected between two drains (24 / 20 / 16 / 12)      */
    int sorted_list_keys;   /* sorted order: neighbors per point the lists of the launch held (128, or what the handle measured;
                               256 or 512 through the word lists)                                                       */
    int sorted_all_large;   /* sorted order: 1 = every point went straight to the wave / workgroup-per-point kernels      */
} kpl_launch_info;
int kpl_get_last_launch(const kpl_detector *h, kpl_launch_info *out);

/* Counters for the algorithmic-bytes model of SURVEY.md 8(d), filled by kpl_collect_stats. */
typedef struct kpl_stats {
    int64_t n_points;        /* points handed in                                              */
    int64_t n_scored;        /* points with finite xyz and normal                             */
    int64_t n_thresholded;   /* scored points with score >= prediction_th                     */
    int64_t sum_kf;          /* sum over scored points of feature-radius neighbors (incl self)*/
    int64_t sum_kn;          /* sum over thresholded points of NMS-radius neighbors           */
    int64_t sum_depth;       /* forest nodes visited (internal + leaf), all trees, all points */
    int64_t n_keypoints;
    int64_t n_cells;
} kpl_stats;

/* ---- lifetime ------------------------------------------------------------------------- */
/* ctor of KeypointLearningDetector (KeypointLearning.h:81-90); `device` = HIP device ordinal. */
int kpl_create(kpl_detector **out, int device);
/* dtor (KeypointLearning.h:93-97).  Waits for the device (kernels of the handle's last *_device calls may still be running). */
void kpl_destroy(kpl_detector *h);
const char *kpl_last_error(const kpl_detector *h);
const char *kpl_status_string(int status);
int kpl_version(void);
/* sha256 (hex) of the sources this library was built from (the .hip and .cpp files of csrc, their headers and this file), as
 * keypoint-learning_amd/build.py computes it: ties a shipped binary to a source tree (tests/test_abi.py). */
const char *kpl_source_hash(void);

/* What the class can do for a view BEFORE compute() -- setInputCloud knows the number of points: sizes the handle's staging
 * arrays (strides of the caller's records; 0 = none), the index tables and the scratch of the scoring path for views of up
 * to n_points, so that the first compute() allocates nothing.  Optional, grow-only, never shrinks; everything it does a
 * first call would do itself.  (Measured on MI355X, profiles/r06_first_call.jsonl: allocation is ~0.3 ms of a first call;
 * what made a first compute() 16-22 ms until round 5 were two stream creations and the first use of the copy paths, which
 * kpl_create now sets up on a thread of its own while the caller loads its forest and cloud.)  No reference counterpart. */
int kpl_reserve(kpl_detector *h, int n_points, size_t xyz_stride, size_t normals_stride);

/* ---- parameters ----------------------------------------------------------------------- */
void kpl_default_params(kpl_params *p);
int kpl_set_params(kpl_detector *h, const kpl_params *p);   /* the setters, hpp:60-113        */
int kpl_get_params(const kpl_detector *h, kpl_params *p);

/* ---- forest: loadForest (hpp:159-176) = cv::ml::RTrees::load ---------------------------- */
/* OpenCV RTrees YAML, plain or gzip (sniffed by magic bytes, not by extension). */
int kpl_load_forest_file(kpl_detector *h, const char *path);
int kpl_load_forest_memory(kpl_detector *h, const void *data, size_t len);
/* Same forest handed over as node arrays (children by global node index, var = -1 for a leaf). */
int kpl_load_forest_arrays(kpl_detector *h, int ntrees, int nnodes, int var_count,
                           const int *root, const int *var, const float *thr,
                           const int *left, const int *right, const double *value);
/* getRoots().size() (hpp:172, :271) and friends. */
int kpl_forest_info(const kpl_detector *h, int *ntrees, int *var_count, int64_t *nnodes,
                    int *max_depth);

/* Host-only forest inspection (no device needed): parses the same YAML / YAML.gz bytes as
 * kpl_load_forest_memory and reports the model, or copies its node arrays out (children by global
 * node index, var = -1 for a leaf; trees in file order, nodes in file (pre-)order).  Arrays may be
 * NULL to query sizes only.  err (optional) receives the parser message on failure. */
typedef struct kpl_forest_summary {
    int ntrees;
    int var_count;
    int64_t nnodes;
    int max_depth;      /* longest root->leaf path counted in nodes */
} kpl_forest_summary;
int kpl_forest_inspect(const void *data, size_t len, kpl_forest_summary *out,
                       char *err, size_t err_cap);
int kpl_forest_export_arrays(const void *data, size_t len, int64_t node_cap, int tree_cap,
                             int *root, int *var, float *thr, int *left, int *right,
                             double *value, char *err, size_t err_cap);

/* ---- compute(): host buffers ------------------------------------------------------------
 * pcl::Keypoint::compute -> initCompute (hpp:116-156) + detectKeypoints (hpp:179-263).
 * xyz / normals point at the first float of element 0; strides are in bytes (>= 12), so
 * pcl::PointXYZ (16 B) and pcl::Normal (32 B) arrays are passed as they are.
 * scores_out (optional, n floats): forest response per input point in INPUT index space, NaN
 * for a point whose xyz or normal is not finite.  kp_idx_out receives the keypoint indices in
 * ascending order (= keypoints_indices_), kp_count their number.  If kp_count > kp_cap the first
 * kp_cap are written and KPL_ERR_CAPACITY is returned. */
int kpl_detect(kpl_detector *h, const void *xyz, size_t xyz_stride,
               const void *normals, size_t normals_stride, int n,
               float *scores_out, int *kp_idx_out, int kp_cap, int *kp_count);

/* The same call for a caller that wants what detectKeypoints() leaves behind -- the keypoint indices
 * and the forest response OF THE KEYPOINTS (the intensity of the output cloud, hpp:258-260) -- and not
 * the response of every point: nothing of size n travels back over PCIe.  kp_scores_out may be NULL. */
int kpl_detect_keypoints(kpl_detector *h, const void *xyz, size_t xyz_stride,
                         const void *normals, size_t normals_stride, int n,
                         int *kp_idx_out, float *kp_scores_out, int kp_cap, int *kp_count);

/* The same for a caller that can put its view into PINNED host memory: kpl_host_staging returns two buffers of
 * the handle (grow-only, valid until the next kpl_host_staging or kpl_destroy) for n points at the given strides
 * (12 = packed xyz / normals; 16 and 32 = pcl::PointXYZ / pcl::Normal records, filled with one memcpy); the caller
 * fills them -- include/KeypointLearning.h does in setInputCloud / setNormals when setHostStaging(true) -- and
 * kpl_detect_keypoints_staged runs compute() on them: the uploads are true DMA copies on a stream of their own, the
 * points first; the index kernels that read only points run while the normals are still in flight.  (Pageable
 * memory, kpl_detect_keypoints: the runtime stages the copies through its own pinned buffer on the calling thread.) */
int kpl_host_staging(kpl_detector *h, int n, size_t xyz_stride, size_t normals_stride, void **xyz, void **normals);
int kpl_detect_keypoints_staged(kpl_detector *h, int *kp_idx_out, float *kp_scores_out, int kp_cap, int *kp_count);

/* computePointsForTrainingFeatures (hpp:299-318): features of the listed points, m x
 * (n_annulus*n_bins) floats, row major.  A row of a point with non-finite xyz is NaN. */
int kpl_compute_features(kpl_detector *h, const void *xyz, size_t xyz_stride,
                         const void *normals, size_t normals_stride, int n,
                         const int *indices, int m, float *features_out);

/* ---- compute(): device-resident buffers, asynchronous -------------------------------------
 * Same path for callers that already hold the view in HBM (the bench, multi-view batches).
 * `stream` is a hipStream_t (NULL = default stream).  kpl_bind_cloud_device only records the
 * pointers.  kpl_compute_device enqueues index build + scoring + NMS + compaction on `stream`;
 * d_kp_count (1 int) and d_kp_idx (kp_cap ints) are written on the device; d_scores optional.
 * kpl_build_index_device / kpl_detect_device are the two halves (initCompute / detectKeypoints)
 * for callers that want them timed apart. */
int kpl_bind_cloud_device(kpl_detector *h, const void *d_xyz, size_t xyz_stride,
                          const void *d_normals, size_t normals_stride, int n);
int kpl_build_index_device(kpl_detector *h, void *stream);
int kpl_detect_device(kpl_detector *h, float *d_scores, int *d_kp_idx, int kp_cap,
                      int *d_kp_count, void *stream);
int kpl_compute_device(kpl_detector *h, float *d_scores, int *d_kp_idx, int kp_cap,
                       int *d_kp_count, void *stream);
int kpl_compute_features_device(kpl_detector *h, const int *d_indices, int m,
                                float *d_features, void *stream);
/* computePointsForTrainingFeatures for a batch of up to 8 bound views (one handle per view, all on one device): the
 * caller loop of /root/reference/src/main_train_detector.cpp:413-446 extracts a few hundred training points from each of
 * many views -- alone, a view is a handful of waves.  The indices of all views are built in one batch of launches, the
 * features in one launch.  Arrays are indexed by view: d_indices[k] = m[k] point indices, d_features[k] = m[k] x
 * (n_annulus*n_bins) floats.  Only enqueues; deferred errors through kpl_sync_status of each handle. */
int kpl_compute_features_batch_device(kpl_detector *const *handles, int count, const int *const *d_indices,
                                      const int *m, float *const *d_features, void *stream);
/* compute() for a batch of up to 8 independent views (one handle per view, each with its cloud
 * bound, forest loaded and parameters set; all on one device).  A single 200 k-point view is only
 * ~3 waves per SIMD on an MI355X; every kernel of the pipeline (index build, scoring, NMS,
 * compaction) is launched ONCE for the whole batch, on `stream`, so the chip is full and the
 * fixed cost of the ~17 launches is shared by the views.  The call only enqueues.  Arrays are indexed by view; d_scores may be NULL (or hold
 * NULLs).  Results per view are exactly those of kpl_compute_device. */
int kpl_compute_batch_device(kpl_detector *const *handles, int count, float *const *d_scores,
                             int *const *d_kp_idx, const int *kp_caps, int *const *d_kp_counts,
                             void *stream);
/* The same batch for a caller that wants what detectKeypoints() leaves behind and nothing else (the device
 * counterpart of kpl_detect_keypoints): per view the keypoint indices, the forest response OF THE KEYPOINTS
 * (d_kp_scores[v][j] belongs to d_kp_idx[v][j]; the array or single entries may be NULL) and the count.  The
 * three outputs of a view may be one packed buffer [count][indices][responses] -- what a multi-GPU caller
 * contributes to its all-gather as it stands (csrc/batch_views_main.cpp). */
int kpl_compute_batch_keypoints_device(kpl_detector *const *handles, int count, int *const *d_kp_idx,
                                       float *const *d_kp_scores, const int *kp_caps, int *const *d_kp_counts,
                                       void *stream);
/* The device entry points never wait for the GPU: the grid descriptor is computed on the device.
 * Some conditions can therefore only be seen afterwards -- a view that needs more than 2^28 grid
 * cells; more cells than the handle's tables currently hold (they start at 8*n + 65536 cells); in
 * sorted-search mode, more neighbor keys of points with large neighborhoods than the handle's key
 * array holds (it starts at 64 keys per point; the reference's default radius on its own test views
 * needs ~2 300); in the two-pass walk of large neighborhoods (kpl_set_feature_walk), more accept words
 * than the handle's word list holds; a device-side consistency check that failed.  In all of them the
 * enqueued call wrote *d_kp_count = -1.  kpl_sync_status waits for `stream` and returns KPL_OK,
 * KPL_ERR_GRID_TOO_LARGE, KPL_ERR_RETRY after growing the table in question (enqueue the call again; a
 * first call may need this more than once, and a handle that switches to the two-pass walk after it has
 * measured the neighborhood size of a view may need it once more then), or KPL_ERR_INTERNAL (call again).
 * It also reads back what the feature kernels measured for the handle's next launch (mean and longest
 * neighborhood: kpl_get_feature_walk).  Growth is ordered by `stream` (hipMallocAsync / hipMemsetAsync /
 * hipFreeAsync): no other stream of the process waits.  kpl_detect / kpl_compute_features do all this
 * internally. */
int kpl_sync_status(kpl_detector *h, void *stream);

/* Per-phase device timing with HIP events recorded on the caller's stream, around the kernels of
 * each phase, for every kpl_build_index_device / kpl_detect_device / kpl_compute_device call made
 * while it is enabled (up to 4096 calls, later ones are not recorded).  kpl_get_timing waits for
 * the recorded events, adds them up and clears the record. */
typedef struct kpl_timing {
    int calls;            /* detect calls summed                                              */
    float index_ms;       /* bounding box + the two-level counting sort of the index build    */
    float score_ms;       /* feature_ms + forest_ms                                           */
    float nms_ms;         /* NMS + flag scan + compaction                                     */
    float feature_ms;     /* the histogram feature kernel (the dominant kernel)               */
    float forest_ms;      /* the forest kernel                                                */
    int walk;             /* KPL_WALK_* the LAST call's feature stage took (-1: sorted order, or no call yet) */
    int lanes_per_point;  /* ... and its lanes per point                                      */
    int accept_words;     /* ... and, for KPL_WALK_LANES, the accept words a point collected between two drains as asked for
                             (24, or 20 / 16 / 12 once the handle has measured a smaller neighborhood; 0 otherwise) */
} kpl_timing;
int kpl_enable_timing(kpl_detector *h, int enable);
int kpl_get_timing(kpl_detector *h, kpl_timing *out);

/* Instrumented pass over the bound cloud (not for timing): fills the counters above. */
int kpl_collect_stats(kpl_detector *h, kpl_stats *out, void *stream);

/* computeCloudResolution (/root/reference/include/impl/point_cloud_utilities.hpp:120-151). */
int kpl_cloud_resolution(kpl_detector *h, const void *xyz, size_t xyz_stride, int n,
                         double *resolution);

/* Grid frame for a view that is a SLAB of a larger cloud (one cloud split over several GPUs; no
 * reference counterpart, the reference is single process).  The canonical neighbor order -- and
 * with it every float sum of the path -- is defined on the grid whose origin is the minimum of
 * the finite points of the view.  A slab that passes the origin of the WHOLE cloud here (and keeps
 * its points in ascending global index order) sees exactly the cells and the order the whole
 * cloud would give, so scores of points whose radius neighborhood lies inside the slab are
 * bit-identical to a single-GPU run.  origin must be <= the slab's own minimum (else
 * KPL_ERR_INVALID_ARG, deferred to kpl_sync_status on the device entry points); NULL restores the
 * automatic origin. */
int kpl_set_grid_origin(kpl_detector *h, const float *origin);

/* pcl::NormalEstimation<PointInT, NormalT> as the reference drives it, on the device:
 *   k_search > 0 (<= 32): setKSearch(k)          -- /root/reference/src/main_test_detector.cpp:162-169
 *                                                   (k = 10, default viewpoint (0,0,0), optional flip :172-179
 *                                                   is the caller's)
 *   k_search <= 0:        setRadiusSearch(radius) -- the detector's own fallback when no normals
 *                                                   were given,
 *                                                   /root/reference/include/impl/KeypointLearning.hpp:125-148
 * Per point: PCA of the neighborhood (the query included), eigenvector of the smallest eigenvalue,
 * flipped towards `viewpoint` (NULL = origin); curvature = lambda_min / trace.  Fewer than 3
 * neighbors or a non-finite point give NaN.  normals_out: 3 floats per point at `normals_stride`
 * bytes (pcl::Normal: 32); curvature_out (may be NULL): 1 float per point at `curvature_stride`
 * bytes (pcl::Normal: &normals[0].curvature, 32).  Arithmetic: DESIGN.md section 2 (double
 * two-pass PCA, cyclic Jacobi) -- PCL's float one-pass covariance and analytic eigen-solver
 * differ from it by ~1e-4 rad ("parity unpinned", PCL absent). */
int kpl_estimate_normals(kpl_detector *h, const void *xyz, size_t xyz_stride, int n, int k_search,
                         double radius_search, const float *viewpoint, void *normals_out,
                         size_t normals_stride, void *curvature_out, size_t curvature_stride);
/* the same on the view bound with kpl_bind_cloud_device (whose normals pointer may be the very
 * buffer that is filled here); asynchronous on `stream`; deferred errors: kpl_sync_status */
int kpl_estimate_normals_device(kpl_detector *h, int k_search, double radius_search,
                                const float *viewpoint, void *d_normals, size_t normals_stride,
                                void *d_curvature, size_t curvature_stride, void *stream);

/* Normals of an ORGANIZED cloud (width x height, row-major) as the detector's own fallback computes them
 * when the caller gave none: include/impl/KeypointLearning.hpp:138-145 = pcl::IntegralImageNormalEstimation,
 * setNormalEstimationMethod(SIMPLE_3D_GRADIENT), setNormalSmoothingSize(5.0) (pass 5.0f), border policy
 * IGNORE, smoothing independent of depth, viewpoint = the cloud's sensor origin (null = (0, 0, 0)).
 * Pixels PCL leaves without a normal (the border of (int)smoothing pixels, non-finite depth, closer than
 * 2 pixels to a depth discontinuity, degenerate gradient) get NaN normals; curvature is NaN everywhere
 * (this method computes none).  PCL 1.8.0 is not part of the reference checkout: restated from its
 * published source, every float in PCL's order ("parity unpinned"). */
int kpl_estimate_normals_organized(kpl_detector *h, const void *xyz, size_t xyz_stride, int width, int height,
                                   float normal_smoothing_size, const float *viewpoint, void *normals_out,
                                   size_t normals_stride, void *curvature_out, size_t curvature_stride);
/* the same on device buffers, asynchronous on `stream` */
int kpl_estimate_normals_organized_device(kpl_detector *h, const void *d_xyz, size_t xyz_stride, int width,
                                          int height, float normal_smoothing_size, const float *viewpoint,
                                          void *d_normals, size_t normals_stride, void *d_curvature,
                                          size_t curvature_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KPL_H */
