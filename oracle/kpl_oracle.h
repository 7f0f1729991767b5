/*
 * kpl_oracle.h -- CPU restatement (ORACLE) of the per-point feature -> forest -> radius-NMS
 * path of pcl::keypoints::KeypointLearningDetector.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py may load it.  The product (libkpl.so, include/kpl.h) never links, loads or calls
 * anything in oracle/.
 *
 * Pinning status: the two soft-assignment functions (findAnnulusPair / findBinPair) are pinned
 * against the reference's own translation unit (oracle/_ref, see oracle/Makefile) and the KAT
 * table in tests/golden/pair_kat.json.  Everything that the reference delegates to PCL/FLANN,
 * Eigen and OpenCV (neighbor enumeration order, dot/norm reduction order, forest predict, YAML)
 * is "PARITY UNPINNED": those libraries are absent from /root/reference and from this image and
 * the reference holds no tests or golden vectors; the normative choices are listed in DESIGN.md.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef KPL_ORACLE_H
#define KPL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Forest as plain node arrays (any node numbering; children by index). */
typedef struct kplo_forest {
    int ntrees;
    int nnodes;
    int var_count;
    const int *root;      /* [ntrees] node index of each tree's root                      */
    const int *var;       /* [nnodes] split variable, or -1 for a leaf                   */
    const float *thr;     /* [nnodes] split threshold (cv::ml DTrees Split::c, a float)  */
    const int *left;      /* [nnodes] child taken when x[var] <= thr                     */
    const int *right;     /* [nnodes] child taken otherwise                              */
    const double *value;  /* [nnodes] node value (class label for a classifier leaf)     */
} kplo_forest;

typedef struct kplo_grid kplo_grid;

/* src/KeypointLearning.cpp:41-65 and :68-92 */
void kplo_find_annulus_pair(int n_annulus, float distance, float support,
                            int *idx, int *pair, float *w);
void kplo_find_bin_pair(int n_bins, float cosine, int *idx, int *pair, float *w);

/* Canonical uniform grid over the finite points of xyz[n*3]; cell edge h = (float)cell_size. */
kplo_grid *kplo_grid_create(const float *xyz, int n, double cell_size);
void kplo_grid_free(kplo_grid *g);
/* grid introspection for tests: dims[3], min[3], h, number of finite points */
void kplo_grid_info(const kplo_grid *g, int *dims, float *mn, float *h, int *nfinite);
/* sorted order (ascending (cell id, original index)) of finite points; out has nfinite ints */
void kplo_grid_sorted_indices(const kplo_grid *g, int *out);

/* Radius search of point `i` (must be finite) in canonical order.  Strict d2 < (float)(r*r).
 * Returns the number of neighbors (including i itself); writes min(count, cap) entries. */
int kplo_radius_search(const kplo_grid *g, const float *xyz, int i, double radius,
                       int *out_idx, float *out_d2, int cap);

/* Neighbor order of the feature loop (include/impl/KeypointLearning.hpp:334-359; element 0 is dropped, :336):
 *   CANONICAL  the grid's canonical order -- stands in for the unknowable FLANN
 *              traversal order of the default pcl::search::KdTree(false);
 *   SORTED     ascending (squared distance, index): what pcl::search::KdTree(true), handed to the inherited
 *              pcl::Keypoint::setSearchMethod, returns (FLANN sorts with DistanceIndex::operator<). */
#define KPLO_ORDER_CANONICAL 0
#define KPLO_ORDER_SORTED 1

/* radius search with sorted results: ascending (d2, index); same set as kplo_radius_search */
int kplo_radius_search_sorted(const kplo_grid *g, const float *xyz, int i, double radius,
                              int *out_idx, float *out_d2, int cap);

/* include/impl/KeypointLearning.hpp:321-376 for each query index; feat_out[m * A*B]. */
void kplo_features(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                   int n_annulus, int n_bins, double r_feat,
                   const int *query, int m, float *feat_out);

void kplo_features_ordered(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                           int n_annulus, int n_bins, double r_feat, int order,
                           const int *query, int m, float *feat_out);

/* cv::ml::RTrees::predict(..., PREDICT_SUM) restated; returns (float)sum of leaf values. */
float kplo_forest_predict_sum(const kplo_forest *f, const float *x, int *depth_sum);

/* include/impl/KeypointLearning.hpp:267-296; scores[n], NaN where point or normal non-finite.
 * n_threads <= 1 runs serial (like the reference); >1 uses that many OpenMP threads. */
void kplo_scores(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                 int n_annulus, int n_bins, double r_feat, const kplo_forest *f,
                 float *scores, int n_threads);

void kplo_scores_ordered(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                         int n_annulus, int n_bins, double r_feat, int order, const kplo_forest *f,
                         float *scores, int n_threads);

/* include/impl/KeypointLearning.hpp:197-261.  Returns number of keypoints written to kp_out
 * (ascending index; capacity n). */
int kplo_nms(const kplo_grid *g, const float *xyz, const float *scores, int n,
             double r_nms, double threshold, int draws_remove, float draws_threshold,
             int *kp_out, int n_threads);

/* Whole path.  non_maxima == 0 mirrors :189-196 (all scoreable points are returned). */
int kplo_detect(const float *xyz, const float *nrm, int n,
                int n_annulus, int n_bins, double r_feat, double r_nms, double threshold,
                int non_maxima, int draws_remove, float draws_threshold,
                const kplo_forest *f, float *scores_out, int *kp_out, int n_threads);

int kplo_detect_ordered(const float *xyz, const float *nrm, int n,
                        int n_annulus, int n_bins, double r_feat, double r_nms, double threshold,
                        int non_maxima, int draws_remove, float draws_threshold, int order,
                        const kplo_forest *f, float *scores_out, int *kp_out, int n_threads);

/* Algorithmic-bytes counters of SURVEY.md 8(d): sum K_f, sum K_n over thresholded points,
 * sum of visited forest nodes. */
void kplo_alg_counters(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                       int n_annulus, int n_bins, double r_feat, double r_nms, double threshold,
                       const kplo_forest *f, int64_t *sum_kf, int64_t *sum_kn,
                       int64_t *sum_depth, int64_t *n_scored, int64_t *n_thresholded);

/* include/impl/point_cloud_utilities.hpp:120-151: mean over finite points of sqrt(2nd-NN d2). */
double kplo_cloud_resolution(const float *xyz, int n);

/* pcl::NormalEstimation restated (src/main_test_detector.cpp:162-169 k-search 10;
 * include/impl/KeypointLearning.hpp:125-148 radius search): k > 0 selects the k-search, else the
 * radius search.  normals_out[3n], curvature_out[n] (may be NULL).  "parity unpinned" (PCL absent). */
void kplo_estimate_normals(const float *xyz, int n, int k, double radius, const float *viewpoint,
                           float *normals_out, float *curvature_out);

/* pcl::IntegralImageNormalEstimation (SIMPLE_3D_GRADIENT, BORDER_POLICY_IGNORE, smoothing independent
 * of depth) restated: what include/impl/KeypointLearning.hpp:138-145 computes for an organized cloud
 * without normals.  xyz[3 * width * height] row-major; normals_out[3 n] (NaN where PCL leaves NaN),
 * curvature_out[n] (all NaN, may be NULL).  "parity unpinned" (PCL absent). */
void kplo_integral_image_normals(const float *xyz, int width, int height, float smoothing_size,
                                 const float *viewpoint, float *normals_out, float *curvature_out);

#ifdef __cplusplus
}
#endif
#endif
