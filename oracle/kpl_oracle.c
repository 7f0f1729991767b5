/*
 * kpl_oracle.c -- CPU restatement (ORACLE) of the KeypointLearningDetector scoring path.
 * TEST INFRASTRUCTURE ONLY -- see kpl_oracle.h for who may use it and for the pinning status
 * ("parity unpinned" for everything the reference delegates to PCL/FLANN, Eigen, OpenCV).
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -std=c11 (see oracle/Makefile).  -ffp-contract=off
 * is part of the specification: the reference arithmetic is separate float mul/add/div/sqrt.
 *
 * Citations are relative to /root/reference.
 */
#include "kpl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KPLO_MAX_CELLS ((int64_t)1 << 28)

/* ------------------------------------------------------------------------------------------
 * Soft assignment.  src/KeypointLearning.cpp:41-65 (annulus) and :68-92 (bin).
 * `abs` there is the float overload on the reference platform (MSVC) -> fabsf here.
 * ---------------------------------------------------------------------------------------- */
void kplo_find_annulus_pair(int n_annulus, float distance, float support,
                            int *idx, int *pair, float *w)
{
    float dim = support / (float)n_annulus;                 /* :43 */
    int k = (int)floorf(distance / dim);                    /* :45 */
    if (k == n_annulus) k--;                                /* :46-47 */
    if (k < 0) k = 0;                 /* :49 assert(0 <= k < n) restated as a clamp: no  */
    if (k > n_annulus - 1) k = n_annulus - 1; /* effect on in-range values (NaN input -> 0) */
    float center = ((float)k * dim) + (dim / 2);            /* :52 */
    float wt = distance - center;                           /* :54 */
    wt /= dim;                                              /* :55 */
    int p = (wt > 0) ? k + 1 : k - 1;                       /* :57 */
    if (p == -1) p = 0;                                     /* :59-60 */
    if (p == n_annulus) p = k;                              /* :61-62 */
    *idx = k;
    *pair = p;
    *w = fabsf(wt);                                         /* :64 */
}

void kplo_find_bin_pair(int n_bins, float cosine, int *idx, int *pair, float *w)
{
    if (cosine < 0) cosine = 0;                             /* :70-71 */
    if (cosine > 2) cosine = 2;                             /* :72-73 */
    float dim = 2 / (float)n_bins;                          /* :75 */
    int k = (int)floorf(cosine / dim);                      /* :76 */
    if (k == n_bins) k--;                                   /* :77-78 */
    if (k < 0) k = 0;                                       /* :80 assert as clamp */
    if (k > n_bins - 1) k = n_bins - 1;
    float center = ((float)k * dim) + (dim / 2);            /* :83 */
    float wt = cosine - center;                             /* :84 */
    wt /= dim;                                              /* :85 */
    int p = (wt > 0) ? k + 1 : k - 1;                       /* :86 */
    if (p == -1) p = 0;                                     /* :87-88 */
    if (p == n_bins) p = k;                                 /* :89-90 */
    *idx = k;
    *pair = p;
    *w = fabsf(wt);                                         /* :91 */
}

/* ------------------------------------------------------------------------------------------
 * Canonical grid (stands in for pcl::search::KdTree -> FLANN, call sites
 * include/impl/KeypointLearning.hpp:213 and :334).  Normative definition:
 *   finite points only; mn = componentwise min; h = (float)cell_size;
 *   cell(v) = clamp((int)floorf((v - mn) / h), 0, dim-1)   (float sub, IEEE float div);
 *   linear id = (cz*ny + cy)*nx + cx;
 *   canonical neighbor order = ascending (linear id of the neighbor's cell, neighbor index).
 * ---------------------------------------------------------------------------------------- */
struct kplo_grid {
    int n, nfinite;
    float mn[3];
    float h;
    int dims[3];
    int64_t ncells;
    int *cell_start; /* [ncells + 1] */
    int *sorted;     /* [nfinite] original indices in canonical storage order */
};

static inline int finite3(const float *p)
{
    return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]);
}

static inline int cell_coord(float v, float mn, float h, int dim)
{
    float t = floorf((v - mn) / h);
    if (!(t >= 0.0f)) return 0;
    if (t >= (float)dim) return dim - 1;
    return (int)t;
}

kplo_grid *kplo_grid_create(const float *xyz, int n, double cell_size)
{
    kplo_grid *g = (kplo_grid *)calloc(1, sizeof(*g));
    if (!g) return NULL;
    g->n = n;
    g->h = (float)cell_size;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int nf = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!finite3(p)) continue;
        ++nf;
        for (int k = 0; k < 3; ++k) {
            if (p[k] < mn[k]) mn[k] = p[k];
            if (p[k] > mx[k]) mx[k] = p[k];
        }
    }
    g->nfinite = nf;
    if (nf == 0 || !(g->h > 0.0f)) {
        g->dims[0] = g->dims[1] = g->dims[2] = 0;
        g->ncells = 0;
        g->cell_start = (int *)calloc(1, sizeof(int));
        g->sorted = (int *)calloc(1, sizeof(int));
        return g;
    }
    int64_t nc = 1;
    for (int k = 0; k < 3; ++k) {
        g->mn[k] = mn[k];
        float t = floorf((mx[k] - mn[k]) / g->h);
        if (!(t < 1.0e9f)) { free(g); return NULL; }
        g->dims[k] = (int)t + 1;
        nc *= g->dims[k];
        if (nc > KPLO_MAX_CELLS) { free(g); return NULL; }
    }
    g->ncells = nc;
    g->cell_start = (int *)calloc((size_t)nc + 1, sizeof(int));
    g->sorted = (int *)malloc(sizeof(int) * (size_t)nf);
    int *cid = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!finite3(p)) { cid[i] = -1; continue; }
        int cx = cell_coord(p[0], g->mn[0], g->h, g->dims[0]);
        int cy = cell_coord(p[1], g->mn[1], g->h, g->dims[1]);
        int cz = cell_coord(p[2], g->mn[2], g->h, g->dims[2]);
        cid[i] = (cz * g->dims[1] + cy) * g->dims[0] + cx;
        g->cell_start[cid[i] + 1]++;
    }
    for (int64_t c = 0; c < nc; ++c) g->cell_start[c + 1] += g->cell_start[c];
    int *fill = (int *)malloc(sizeof(int) * (size_t)nc);
    memcpy(fill, g->cell_start, sizeof(int) * (size_t)nc);
    for (int i = 0; i < n; ++i)           /* ascending i => ascending index inside a cell */
        if (cid[i] >= 0) g->sorted[fill[cid[i]]++] = i;
    free(fill);
    free(cid);
    return g;
}

void kplo_grid_free(kplo_grid *g)
{
    if (!g) return;
    free(g->cell_start);
    free(g->sorted);
    free(g);
}

void kplo_grid_info(const kplo_grid *g, int *dims, float *mn, float *h, int *nfinite)
{
    for (int k = 0; k < 3; ++k) { dims[k] = g->dims[k]; mn[k] = g->mn[k]; }
    *h = g->h;
    *nfinite = g->nfinite;
}

void kplo_grid_sorted_indices(const kplo_grid *g, int *out)
{
    memcpy(out, g->sorted, sizeof(int) * (size_t)g->nfinite);
}

/* FLANN L2_Simple<float> restated: d = dx*dx; d += dy*dy; d += dz*dz (float). */
static inline float dist2(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float d = dx * dx;
    d += dy * dy;
    d += dz * dz;
    return d;
}

typedef struct {
    int lo[3], hi[3];
    float r2;
} search_box;

/* KdTreeFLANN::radiusSearch: r2 = (float)(radius*radius), product in double; strict d2 < r2.
 * The cell range is found with the same monotonic cell function applied to p -+ rr with
 * rr slightly above r, so every accepted point is covered whatever the rounding. */
static inline void make_box(const kplo_grid *g, const float *p, double radius, search_box *b)
{
    float rr = (float)(radius * (1.0 + 1.0 / 1024.0));
    b->r2 = (float)(radius * radius);
    for (int k = 0; k < 3; ++k) {
        b->lo[k] = cell_coord(p[k] - rr, g->mn[k], g->h, g->dims[k]);
        b->hi[k] = cell_coord(p[k] + rr, g->mn[k], g->h, g->dims[k]);
    }
}

#define FOR_EACH_CANDIDATE(g, b, J, ...)                                                  \
    for (int cz_ = (b).lo[2]; cz_ <= (b).hi[2]; ++cz_)                                      \
        for (int cy_ = (b).lo[1]; cy_ <= (b).hi[1]; ++cy_) {                                \
            int row_ = (cz_ * (g)->dims[1] + cy_) * (g)->dims[0];                           \
            int s0_ = (g)->cell_start[row_ + (b).lo[0]];                                    \
            int s1_ = (g)->cell_start[row_ + (b).hi[0] + 1];                                \
            for (int s_ = s0_; s_ < s1_; ++s_) {                                            \
                int J = (g)->sorted[s_];                                                    \
                __VA_ARGS__                                                                 \
            }                                                                               \
        }

int kplo_radius_search(const kplo_grid *g, const float *xyz, int i, double radius,
                       int *out_idx, float *out_d2, int cap)
{
    const float *p = xyz + 3 * (size_t)i;
    if (!finite3(p) || g->ncells == 0) return 0;
    search_box b;
    make_box(g, p, radius, &b);
    int cnt = 0;
    FOR_EACH_CANDIDATE(g, b, j, {
        float d2 = dist2(p, xyz + 3 * (size_t)j);
        if (d2 < b.r2) {
            if (cnt < cap) {
                if (out_idx) out_idx[cnt] = j;
                if (out_d2) out_d2[cnt] = d2;
            }
            ++cnt;
        }
    })
    return cnt;
}

/* The same search with SORTED results: what the caller of the inherited
 * pcl::Keypoint::setSearchMethod gets from a pcl::search::KdTree constructed with sorted = true
 * (include/KeypointLearning.h:56 inherits the setter; call sites include/impl/KeypointLearning.hpp:213,
 * :334).  KdTreeFLANN::radiusSearch then asks FLANN for sorted results, and FLANN (absent from
 * /root/reference; published source, flann/util/result_set.h) sorts its RadiusResultSet with
 * DistanceIndex::operator<: ascending distance, ties by ascending index.  The set is the same as in
 * canonical order (strict d2 < r2, dist2() above); only the order differs.  This is the one neighbor
 * order that is fully defined without FLANN's tree layout. */
typedef struct {
    float d2;
    int idx;
} sorted_item;

static int sorted_item_cmp(const void *a, const void *b)
{
    const sorted_item *x = (const sorted_item *)a, *y = (const sorted_item *)b;
    if (x->d2 < y->d2) return -1;
    if (x->d2 > y->d2) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* all neighbors of point i, ascending (d2, index); *items is grown with realloc */
static int sorted_neighbors(const kplo_grid *g, const float *xyz, int i, double radius,
                            sorted_item **items, int *cap)
{
    const float *p = xyz + 3 * (size_t)i;
    search_box b;
    make_box(g, p, radius, &b);
    int cnt = 0;
    FOR_EACH_CANDIDATE(g, b, j, {
        float d2 = dist2(p, xyz + 3 * (size_t)j);
        if (d2 < b.r2) {
            if (cnt == *cap) {
                *cap = *cap ? *cap * 2 : 256;
                *items = (sorted_item *)realloc(*items, sizeof(sorted_item) * (size_t)*cap);
            }
            (*items)[cnt].d2 = d2;
            (*items)[cnt].idx = j;
            ++cnt;
        }
    })
    qsort(*items, (size_t)cnt, sizeof(sorted_item), sorted_item_cmp);
    return cnt;
}

int kplo_radius_search_sorted(const kplo_grid *g, const float *xyz, int i, double radius,
                              int *out_idx, float *out_d2, int cap)
{
    const float *p = xyz + 3 * (size_t)i;
    if (!finite3(p) || g->ncells == 0) return 0;
    sorted_item *items = NULL;
    int icap = 0;
    const int cnt = sorted_neighbors(g, xyz, i, radius, &items, &icap);
    for (int k = 0; k < cnt && k < cap; ++k) {
        if (out_idx) out_idx[k] = items[k].idx;
        if (out_d2) out_d2[k] = items[k].d2;
    }
    free(items);
    return cnt;
}

/* what one neighbor adds to the histogram, include/impl/KeypointLearning.hpp:338-355 */
static inline void add_neighbor(const float *np, const float *nq, float d2, int A, int B, float support, float *H)
{
    if (!finite3(nq)) return;                  /* :338 */
    /* :342  Eigen Vector3f::dot, unrolled as x + (y + z) */
    float dot = np[0] * nq[0] + (np[1] * nq[1] + np[2] * nq[2]);
    float cosine = 1 - dot;
    int a, ap, bi, bp;
    float aw, bw;
    kplo_find_annulus_pair(A, sqrtf(d2), support, &a, &ap, &aw);   /* :345 */
    kplo_find_bin_pair(B, cosine, &bi, &bp, &bw);                  /* :348 */
    H[a * B + bi] += ((1 - bw) * (1 - aw));                        /* :350 */
    H[a * B + bp] += ((bw) * (1 - aw));                            /* :351 */
    H[ap * B + bi] += ((1 - bw) * (aw));                           /* :354 */
    H[ap * B + bp] += ((bw) * (aw));                               /* :355 */
}

/* ------------------------------------------------------------------------------------------
 * computePointFeatures, include/impl/KeypointLearning.hpp:321-376.
 * H is A x B row-major (the reference's Eigen matrix is column-major; only (row, col)
 * addressing matters).  Returns K_f (neighbors found, including the dropped first one).
 * order: KPLO_ORDER_CANONICAL = neighbors in the grid's canonical order, KPLO_ORDER_SORTED = ascending
 * (d2, index).  Either way the loop starts at the SECOND neighbor (:336).
 * ---------------------------------------------------------------------------------------- */
static int point_features(const kplo_grid *g, const float *xyz, const float *nrm, int i,
                          int A, int B, double r_feat, int order, float *H /* A*B */)
{
    const int F = A * B;
    for (int c = 0; c < F; ++c) H[c] = 0.0f;                       /* :325 */
    const float *p = xyz + 3 * (size_t)i;
    const float *np = nrm + 3 * (size_t)i;                         /* :332 */
    const float support = (float)r_feat;      /* double search_radius_ -> float param, :345 */
    search_box b;
    make_box(g, p, r_feat, &b);
    int seen = 0;
    if (order == KPLO_ORDER_SORTED) {
        sorted_item *items = NULL;
        int icap = 0;
        seen = sorted_neighbors(g, xyz, i, r_feat, &items, &icap);            /* :334 */
        for (int k = 1; k < seen; ++k)                                         /* :336 */
            add_neighbor(np, nrm + 3 * (size_t)items[k].idx, items[k].d2, A, B, support, H);
        free(items);
    } else {
    FOR_EACH_CANDIDATE(g, b, j, {
        const float *q = xyz + 3 * (size_t)j;
        float d2 = dist2(p, q);
        if (d2 < b.r2) {
            if (seen++ == 0) continue;                 /* :336 loop starts at neigh_indx = 1 */
            add_neighbor(np, nrm + 3 * (size_t)j, d2, A, B, support, H);
        }
    })
    }
    for (int a = 0; a < A; ++a) {                                          /* :360-370 */
        float s = 0.0f;
        for (int k = 0; k < B; ++k) s += H[a * B + k] * H[a * B + k];
        float nr = sqrtf(s);
        if (nr > 0)
            for (int k = 0; k < B; ++k) H[a * B + k] = H[a * B + k] / nr;
    }
    return seen;
}

void kplo_features(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                   int n_annulus, int n_bins, double r_feat,
                   const int *query, int m, float *feat_out)
{
    kplo_features_ordered(g, xyz, nrm, n, n_annulus, n_bins, r_feat, KPLO_ORDER_CANONICAL, query, m, feat_out);
}

void kplo_features_ordered(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                           int n_annulus, int n_bins, double r_feat, int order,
                           const int *query, int m, float *feat_out)
{
    (void)n;
    const int F = n_annulus * n_bins;
    for (int q = 0; q < m; ++q) {
        int i = query[q];
        float *out = feat_out + (size_t)q * F;
        if (!finite3(xyz + 3 * (size_t)i)) {   /* the reference would search a NaN query: undefined */
            for (int c = 0; c < F; ++c) out[c] = NAN;
            continue;
        }
        point_features(g, xyz, nrm, i, n_annulus, n_bins, r_feat, order, out);
    }
}

/* ------------------------------------------------------------------------------------------
 * cv::ml::RTrees::predict(feat, result, PREDICT_SUM) restated (OpenCV 3.2 DTreesImpl::
 * predictTrees): per tree walk "val <= c ? left : right", sum leaf values in double,
 * return (float)sum.  Call site include/impl/KeypointLearning.hpp:281.
 * ---------------------------------------------------------------------------------------- */
float kplo_forest_predict_sum(const kplo_forest *f, const float *x, int *depth_sum)
{
    double sum = 0.0;
    int depth = 0;
    for (int t = 0; t < f->ntrees; ++t) {
        int nd = f->root[t];
        while (f->var[nd] >= 0) {
            float val = x[f->var[nd]];
            nd = (val <= f->thr[nd]) ? f->left[nd] : f->right[nd];
            ++depth;
        }
        ++depth; /* the leaf itself is a visited node */
        sum += f->value[nd];
    }
    if (depth_sum) *depth_sum = depth;
    return (float)sum;
}

/* runForest, include/impl/KeypointLearning.hpp:267-296.  Index space = input index space;
 * a non-scoreable point gets NaN instead of being compacted away (documented divergence,
 * identical for all-finite clouds). */
void kplo_scores(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                 int n_annulus, int n_bins, double r_feat, const kplo_forest *f,
                 float *scores, int n_threads)
{
    kplo_scores_ordered(g, xyz, nrm, n, n_annulus, n_bins, r_feat, KPLO_ORDER_CANONICAL, f, scores, n_threads);
}

void kplo_scores_ordered(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                         int n_annulus, int n_bins, double r_feat, int order, const kplo_forest *f,
                         float *scores, int n_threads)
{
    const int F = n_annulus * n_bins;
    const int forest_size = f->ntrees;                                   /* :271 */
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        float *H = (float *)malloc(sizeof(float) * (size_t)F);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int i = 0; i < n; ++i) {
            if (!(finite3(xyz + 3 * (size_t)i) && finite3(nrm + 3 * (size_t)i))) { /* :277 */
                scores[i] = NAN;
                continue;
            }
            point_features(g, xyz, nrm, i, n_annulus, n_bins, r_feat, order, H);  /* :279 */
            const float sum = kplo_forest_predict_sum(f, H, NULL);         /* :281 */
            scores[i] = 1 - (sum / (forest_size * 1.0f));                  /* :287 */
        }
        free(H);
    }
    (void)n_threads;
}

/* ------------------------------------------------------------------------------------------
 * Non-maxima suppression, include/impl/KeypointLearning.hpp:197-261.
 * ---------------------------------------------------------------------------------------- */
static int nms_point(const kplo_grid *g, const float *xyz, const float *scores, int idx,
                     double r_nms, int *has_draw_out, int **draws, int *ndraws, int *cap_draws)
{
    const float *p = xyz + 3 * (size_t)idx;
    search_box b;
    make_box(g, p, r_nms, &b);                                            /* :213 */
    int is_maxima = 1, has_draw = 0;
    const float si = scores[idx];
    if (ndraws) *ndraws = 0;
    FOR_EACH_CANDIDATE(g, b, j, {
        if (dist2(p, xyz + 3 * (size_t)j) < b.r2) {
            if (si < scores[j]) {                                         /* :219 */
                is_maxima = 0;
                goto done;                                                /* :222 break */
            } else if (si == scores[j]) {                                 /* :224 */
                if (idx != j) {
                    has_draw = 1;
                    if (draws) {
                        if (*ndraws == *cap_draws) {
                            *cap_draws = *cap_draws ? *cap_draws * 2 : 64;
                            *draws = (int *)realloc(*draws, sizeof(int) * (size_t)*cap_draws);
                        }
                        (*draws)[(*ndraws)++] = j;                        /* :227 */
                    }
                }
            }
        }
    })
done:
    *has_draw_out = has_draw;
    return is_maxima;
}

int kplo_nms(const kplo_grid *g, const float *xyz, const float *scores, int n,
             double r_nms, double threshold, int draws_remove, float draws_threshold,
             int *kp_out, int n_threads)
{
    int count = 0;
    if (!draws_remove) {
        /* predicate form: order independent, so it may run in parallel */
        unsigned char *flag = (unsigned char *)calloc((size_t)n + 1, 1);
#ifdef _OPENMP
        if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads)
#endif
        for (int idx = 0; idx < n; ++idx) {
            if (!finite3(xyz + 3 * (size_t)idx) || !isfinite(scores[idx]) ||
                (double)scores[idx] < threshold)                           /* :205-208 */
                continue;
            int hd;
            if (nms_point(g, xyz, scores, idx, r_nms, &hd, NULL, NULL, NULL)) flag[idx] = 1;
        }
        for (int idx = 0; idx < n; ++idx)
            if (flag[idx]) kp_out[count++] = idx;                          /* :252-253 */
        free(flag);
        (void)n_threads;
        return count;
    }
    /* draws_remove: order-dependent greedy pass, serial like the reference (:231-250).
     * skipList membership (std::find, :234) is kept as a flag array. */
    unsigned char *skip = (unsigned char *)calloc((size_t)n + 1, 1);
    int *draws = NULL, ndraws = 0, cap = 0;
    for (int idx = 0; idx < n; ++idx) {
        if (!finite3(xyz + 3 * (size_t)idx) || !isfinite(scores[idx]) ||
            (double)scores[idx] < threshold)
            continue;
        int has_draw;
        int is_max = nms_point(g, xyz, scores, idx, r_nms, &has_draw, &draws, &ndraws, &cap);
        if (!is_max) continue;
        if (has_draw) {
            if (!skip[idx]) {                                              /* :234 */
                int survive = 0;
                const float *p = xyz + 3 * (size_t)idx;
                for (int k = 0; k < ndraws; ++k) {
                    const float *q = xyz + 3 * (size_t)draws[k];
                    /* :239 (a - b).norm(): Eigen squaredNorm of a fixed 3-vector,
                     * unrolled x*x + (y*y + z*z), then sqrt */
                    float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                    float distance = sqrtf(dx * dx + (dy * dy + dz * dz));
                    if (distance < draws_threshold) {                      /* :240 */
                        survive = 1;
                        skip[draws[k]] = 1;                                /* :242 */
                    }
                }
                if (survive) kp_out[count++] = idx;                        /* :245-248 */
            }
        } else {
            kp_out[count++] = idx;                                         /* :252-253 */
        }
    }
    free(draws);
    free(skip);
    return count;
}

int kplo_detect(const float *xyz, const float *nrm, int n,
                int n_annulus, int n_bins, double r_feat, double r_nms, double threshold,
                int non_maxima, int draws_remove, float draws_threshold,
                const kplo_forest *f, float *scores_out, int *kp_out, int n_threads)
{
    return kplo_detect_ordered(xyz, nrm, n, n_annulus, n_bins, r_feat, r_nms, threshold, non_maxima, draws_remove,
                               draws_threshold, KPLO_ORDER_CANONICAL, f, scores_out, kp_out, n_threads);
}

int kplo_detect_ordered(const float *xyz, const float *nrm, int n,
                        int n_annulus, int n_bins, double r_feat, double r_nms, double threshold,
                        int non_maxima, int draws_remove, float draws_threshold, int order,
                        const kplo_forest *f, float *scores_out, int *kp_out, int n_threads)
{
    kplo_grid *g = kplo_grid_create(xyz, n, r_feat);
    if (!g) return -1;
    float *scores = scores_out ? scores_out : (float *)malloc(sizeof(float) * (size_t)(n + 1));
    kplo_scores_ordered(g, xyz, nrm, n, n_annulus, n_bins, r_feat, order, f, scores, n_threads);
    int count = 0;
    if (!non_maxima) {
        /* :189-196: every response point is a keypoint.  In input index space that is every
         * scoreable point. */
        for (int i = 0; i < n; ++i)
            if (!isnan(scores[i])) kp_out[count++] = i;
    } else {
        count = kplo_nms(g, xyz, scores, n, r_nms, threshold, draws_remove, draws_threshold,
                         kp_out, n_threads);
    }
    if (!scores_out) free(scores);
    kplo_grid_free(g);
    return count;
}

void kplo_alg_counters(const kplo_grid *g, const float *xyz, const float *nrm, int n,
                       int n_annulus, int n_bins, double r_feat, double r_nms, double threshold,
                       const kplo_forest *f, int64_t *sum_kf, int64_t *sum_kn,
                       int64_t *sum_depth, int64_t *n_scored, int64_t *n_thresholded)
{
    const int F = n_annulus * n_bins;
    float *H = (float *)malloc(sizeof(float) * (size_t)F);
    int64_t kf = 0, kn = 0, dp = 0, ns = 0, nt = 0;
    for (int i = 0; i < n; ++i) {
        if (!(finite3(xyz + 3 * (size_t)i) && finite3(nrm + 3 * (size_t)i))) continue;
        kf += point_features(g, xyz, nrm, i, n_annulus, n_bins, r_feat, KPLO_ORDER_CANONICAL, H);
        int d;
        float sum = kplo_forest_predict_sum(f, H, &d);
        dp += d;
        ++ns;
        float score = 1 - (sum / (f->ntrees * 1.0f));
        if (!((double)score < threshold)) {
            ++nt;
            kn += kplo_radius_search(g, xyz, i, r_nms, NULL, NULL, 0);
        }
    }
    free(H);
    *sum_kf = kf; *sum_kn = kn; *sum_depth = dp; *n_scored = ns; *n_thresholded = nt;
}

/* ------------------------------------------------------------------------------------------
 * computeCloudResolution, include/impl/point_cloud_utilities.hpp:120-151: mean over points
 * with finite x of sqrt(second smallest squared distance), double accumulator.
 * ---------------------------------------------------------------------------------------- */
/* grid for nearest-neighbor queries without a radius: about two points per cell on a
 * surface-like cloud (the results do not depend on the cell size, only the work does) */
static kplo_grid *auto_grid(const float *xyz, int n)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int nf = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!finite3(p)) continue;
        ++nf;
        for (int k = 0; k < 3; ++k) {
            if (p[k] < mn[k]) mn[k] = p[k];
            if (p[k] > mx[k]) mx[k] = p[k];
        }
    }
    if (nf == 0) return kplo_grid_create(xyz, n, 1.0);
    double ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    double e0 = ext[0], e1 = ext[1], e2 = ext[2], t;
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    if (e1 < e2) { t = e1; e1 = e2; e2 = t; }
    if (e0 < e1) { t = e0; e0 = e1; e1 = t; }
    double cell = sqrt((e0 * (e1 > 0 ? e1 : e0)) / (double)nf * 2.0);
    if (!(cell > 0)) cell = 1.0;
    kplo_grid *g = NULL;
    while (!(g = kplo_grid_create(xyz, n, cell))) cell *= 2.0;
    return g;
}

double kplo_cloud_resolution(const float *xyz, int n)
{
    kplo_grid *g = auto_grid(xyz, n);
    if (g->nfinite < 2) { kplo_grid_free(g); return 0.0; }
    double res = 0.0;
    int n_points = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        if (!finite3(p)) continue;
        int c[3];
        for (int k = 0; k < 3; ++k) c[k] = cell_coord(p[k], g->mn[k], g->h, g->dims[k]);
        float best0 = INFINITY, best1 = INFINITY;
        int maxring = g->dims[0] > g->dims[1] ? g->dims[0] : g->dims[1];
        if (g->dims[2] > maxring) maxring = g->dims[2];
        for (int ring = 1; ring <= maxring; ++ring) {
            search_box b;
            b.r2 = 0;
            for (int k = 0; k < 3; ++k) {
                b.lo[k] = c[k] - ring < 0 ? 0 : c[k] - ring;
                b.hi[k] = c[k] + ring >= g->dims[k] ? g->dims[k] - 1 : c[k] + ring;
            }
            best0 = best1 = INFINITY;
            FOR_EACH_CANDIDATE(g, b, j, {
                float d2 = dist2(p, xyz + 3 * (size_t)j);
                if (d2 < best0) { best1 = best0; best0 = d2; }
                else if (d2 < best1) best1 = d2;
            })
            double safe = ring * (double)g->h * 0.999;
            if (isfinite(best1) && (double)sqrtf(best1) <= safe) break;
            if (b.lo[0] == 0 && b.lo[1] == 0 && b.lo[2] == 0 && b.hi[0] == g->dims[0] - 1 &&
                b.hi[1] == g->dims[1] - 1 && b.hi[2] == g->dims[2] - 1)
                break;
        }
        if (isfinite(best1)) {
            res += sqrtf(best1);                                          /* :141 */
            ++n_points;
        }
    }
    kplo_grid_free(g);
    if (n_points != 0) res /= n_points;                                   /* :145-148 */
    return res;
}

/* ------------------------------------------------------------------------------------------
 * Normal estimation: pcl::NormalEstimation as the reference uses it --
 *   src/main_test_detector.cpp:162-169   setKSearch(10), viewpoint (0,0,0)
 *   include/impl/KeypointLearning.hpp:125-148   setRadiusSearch(search_radius_) when the caller
 *                                               gave no normals (unorganized cloud)
 * PCL is absent from the image, so this is a restatement of its published algorithm
 * (computePointNormal: PCA of the neighborhood, eigenvector of the smallest eigenvalue,
 * flipNormalTowardsViewpoint, curvature = lambda_min / trace) with the arithmetic fixed as
 * follows ("parity unpinned": PCL accumulates the covariance in float in one pass and solves the
 * eigenproblem analytically in float; results agree to ~1e-4 rad, not bitwise):
 *   neighbors   k-search: the k smallest (d2, index), d2 as in dist2(), the query included,
 *               in that order; radius search: kplo_radius_search order (grid cell = radius)
 *   fewer than 3 neighbors, or a non-finite query: NaN normal and curvature
 *   mean, covariance: double, two passes, sequential sums in neighbor order
 *   eigenvectors: cyclic Jacobi in double (pairs (0,1),(0,2),(1,2), at most 24 sweeps, stops
 *               when the off-diagonal energy is below 1e-36 of the diagonal energy);
 *               smallest diagonal entry, first one on ties
 *   flip        if n . (viewpoint - p) < 0 (double)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float d2;
    int idx;
} knn_item;

static int knn_less(float d2, int idx, const knn_item *b)
{
    return d2 < b->d2 || (d2 == b->d2 && idx < b->idx);
}

/* the k nearest finite points of point i (itself included), ascending (d2, index) */
static int knn_search(const kplo_grid *g, const float *xyz, int i, int k, knn_item *best)
{
    const float *p = xyz + 3 * (size_t)i;
    int c[3], m = 0;
    for (int a = 0; a < 3; ++a) c[a] = cell_coord(p[a], g->mn[a], g->h, g->dims[a]);
    int maxring = g->dims[0] > g->dims[1] ? g->dims[0] : g->dims[1];
    if (g->dims[2] > maxring) maxring = g->dims[2];
    for (int ring = 1; ring <= maxring; ++ring) {
        search_box b;
        b.r2 = 0;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = c[a] - ring < 0 ? 0 : c[a] - ring;
            b.hi[a] = c[a] + ring >= g->dims[a] ? g->dims[a] - 1 : c[a] + ring;
        }
        m = 0;
        FOR_EACH_CANDIDATE(g, b, j, {
            float d2 = dist2(p, xyz + 3 * (size_t)j);
            if (m < k || knn_less(d2, j, &best[m - 1])) {
                int pos = m < k ? m++ : k - 1;
                while (pos > 0 && knn_less(d2, j, &best[pos - 1])) {
                    best[pos] = best[pos - 1];
                    --pos;
                }
                best[pos].d2 = d2;
                best[pos].idx = j;
            }
        })
        /* every point outside the block is at least ring*h away (0.999: slack for float cell edges) */
        if (m == k && (double)sqrtf(best[k - 1].d2) <= ring * (double)g->h * 0.999) break;
        if (b.lo[0] == 0 && b.lo[1] == 0 && b.lo[2] == 0 && b.hi[0] == g->dims[0] - 1 &&
            b.hi[1] == g->dims[1] - 1 && b.hi[2] == g->dims[2] - 1)
            break;
    }
    return m;
}

/* eigenvector of the smallest eigenvalue of the symmetric 3x3 matrix a (destroyed) */
static void jacobi_smallest(double a[3][3], double v[3], double *lambda, double *trace)
{
    double e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 24; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (!(off > 1e-36 * dg)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[p][q];
                if (apq == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double ekp = e[k][p], ekq = e[k][q];
                    e[k][p] = c * ekp - s * ekq;
                    e[k][q] = s * ekp + c * ekq;
                }
            }
    }
    int m = 0;
    for (int k = 1; k < 3; ++k)
        if (a[k][k] < a[m][m]) m = k;
    for (int k = 0; k < 3; ++k) v[k] = e[k][m];
    *lambda = a[m][m];
    *trace = a[0][0] + a[1][1] + a[2][2];
}

static void normal_from_neighbors(const float *xyz, const float *p, const int *idx, int m,
                                  const float *vp, float *out, float *curv)
{
    if (m < 3) {
        out[0] = out[1] = out[2] = NAN;
        if (curv) *curv = NAN;
        return;
    }
    double sum[3] = {0, 0, 0};
    for (int t = 0; t < m; ++t)
        for (int a = 0; a < 3; ++a) sum[a] += (double)xyz[3 * (size_t)idx[t] + a];
    double mean[3];
    for (int a = 0; a < 3; ++a) mean[a] = sum[a] / (double)m;
    double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int t = 0; t < m; ++t) {
        const float *q = xyz + 3 * (size_t)idx[t];
        const double dx = (double)q[0] - mean[0], dy = (double)q[1] - mean[1], dz = (double)q[2] - mean[2];
        cov[0][0] += dx * dx; cov[0][1] += dx * dy; cov[0][2] += dx * dz;
        cov[1][1] += dy * dy; cov[1][2] += dy * dz; cov[2][2] += dz * dz;
    }
    cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
    double v[3], lambda, trace;
    jacobi_smallest(cov, v, &lambda, &trace);
    const double dot = v[0] * ((double)vp[0] - (double)p[0]) + v[1] * ((double)vp[1] - (double)p[1]) +
                       v[2] * ((double)vp[2] - (double)p[2]);
    if (dot < 0.0) { v[0] = -v[0]; v[1] = -v[1]; v[2] = -v[2]; }
    out[0] = (float)v[0]; out[1] = (float)v[1]; out[2] = (float)v[2];
    if (curv) *curv = trace != 0.0 ? (float)fabs(lambda / trace) : 0.0f;
}

void kplo_estimate_normals(const float *xyz, int n, int k, double radius, const float *viewpoint,
                           float *normals_out, float *curvature_out)
{
    const int use_k = k > 0;
    kplo_grid *g = use_k ? auto_grid(xyz, n) : kplo_grid_create(xyz, n, radius);
    int cap = use_k ? k : 64;
    int *idx = (int *)malloc(sizeof(int) * (size_t)cap);
    knn_item *best = use_k ? (knn_item *)malloc(sizeof(knn_item) * (size_t)k) : NULL;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + 3 * (size_t)i;
        float *out = normals_out + 3 * (size_t)i;
        float *cv = curvature_out ? curvature_out + i : NULL;
        int m = 0;
        if (g && finite3(p) && g->ncells > 0) {
            if (use_k) {
                m = knn_search(g, xyz, i, k, best);
                for (int t = 0; t < m; ++t) idx[t] = best[t].idx;
            } else {
                m = kplo_radius_search(g, xyz, i, radius, idx, NULL, cap);
                if (m > cap) {
                    cap = m * 2;
                    idx = (int *)realloc(idx, sizeof(int) * (size_t)cap);
                    m = kplo_radius_search(g, xyz, i, radius, idx, NULL, cap);
                }
            }
        }
        normal_from_neighbors(xyz, p, idx, m, viewpoint, out, cv);
    }
    free(idx);
    free(best);
    kplo_grid_free(g);
}

/* ---------------------------------------------------------------------------------------------
 * pcl::IntegralImageNormalEstimation as the detector's fallback drives it on an ORGANIZED cloud
 * (include/impl/KeypointLearning.hpp:138-145): setNormalEstimationMethod(SIMPLE_3D_GRADIENT),
 * setInputCloud, setNormalSmoothingSize(5.0), compute().  The algorithm lives in PCL 1.8.0
 * (README.md:66), absent from /root/reference: features/include/pcl/features/impl/
 * integral_image_normal.hpp (computeFeature, computeFeatureFull with BORDER_POLICY_IGNORE and
 * depth-independent smoothing, computePointNormal's SIMPLE_3D_GRADIENT branch) and
 * integral_image2D.hpp (first-order integral image, double sums).  Restated here from the
 * published source, loop by loop, "parity unpinned":
 *   1. depth-change map: a pixel and its right / lower neighbor are marked where their depths differ
 *      by more than max_depth_change_factor (20.0f * 0.001f) * (|z| + 1) * 2 or either is not finite
 *   2. distance map: 0 on marked pixels, width + height elsewhere, then the two chamfer passes
 *      (1.0 straight, 1.4 diagonal) with PCL's loop bounds -- the first pass runs ci up to
 *      width - 1 and reads previous_row[ci + 1], i.e. the first pixel of the current row at the
 *      right edge; the second reads next_row[ci - 1] at ci = 0.  Neither reaches a pixel that is
 *      kept: the 5-pixel border is set to NaN and values >= 5 are clipped to 5
 *   3. first-order integral image of (x, y, z) in double, points whose x + (y + z) (float) is not
 *      finite counted as zero: cur[c + 1] = prev[c + 1] + cur[c] - prev[c] (+ point)
 *   4. per kept pixel with finite z: s = min(distance, 5); s > 2 -> rectangle (int)s x (int)s:
 *      gx = column sum at x + w/2 minus column sum at x - w/2, gy = row sum at y + h/2 minus row sum
 *      at y - h/2 (both from y - h/2 resp. x - w/2), n = gy x gx normalised in double, cast to float,
 *      flipped towards the viewpoint (sensor origin = 0 unless given); curvature = NaN.
 * Eigen detail that cannot be pinned: n /= sqrt(len) is a true division here (Eigen 3.3; 3.2
 * multiplies by the reciprocal) -- at most one ulp of the double before the cast to float.
 * ------------------------------------------------------------------------------------------- */
static void ii_sum(const double *ii, int W, int sx, int sy, int w, int h, double out[3])
{
    const size_t ul = (size_t)sy * (size_t)(W + 1) + (size_t)sx, ur = ul + (size_t)w;
    const size_t ll = (size_t)(sy + h) * (size_t)(W + 1) + (size_t)sx, lr = ll + (size_t)w;
    for (int a = 0; a < 3; ++a) out[a] = ((ii[3 * lr + a] + ii[3 * ul + a]) - ii[3 * ur + a]) - ii[3 * ll + a];
}

void kplo_integral_image_normals(const float *xyz, int width, int height, float smoothing_size,
                                 const float *viewpoint, float *normals_out, float *curvature_out)
{
    const int W = width, H = height;
    const size_t n = (size_t)(W > 0 ? W : 0) * (size_t)(H > 0 ? H : 0);
    const float bad = NAN;
    for (size_t i = 0; i < n; ++i) {
        normals_out[3 * i] = normals_out[3 * i + 1] = normals_out[3 * i + 2] = bad;
        if (curvature_out) curvature_out[i] = bad;
    }
    const int border = (int)smoothing_size;
    if (n == 0 || border < 0 || W <= 2 * border || H <= 2 * border) return;
    const float vp[3] = {viewpoint ? viewpoint[0] : 0.0f, viewpoint ? viewpoint[1] : 0.0f, viewpoint ? viewpoint[2] : 0.0f};
    /* 1. depth-change map */
    unsigned char *change = (unsigned char *)malloc(n);
    memset(change, 255, n);
    const float factor = 20.0f * 0.001f;
    for (int ri = 0; ri < H - 1; ++ri)
        for (int ci = 0; ci < W - 1; ++ci) {
            const size_t index = (size_t)ri * (size_t)W + (size_t)ci;
            const float depth = xyz[3 * index + 2], depthR = xyz[3 * (index + 1) + 2], depthD = xyz[3 * (index + (size_t)W) + 2];
            const float limit = (factor * (fabsf(depth) + 1.0f) * 2.0f);
            if (fabs(depth - depthR) > limit || !isfinite(depth) || !isfinite(depthR)) {
                change[index] = 0;
                change[index + 1] = 0;
            }
            if (fabs(depth - depthD) > limit || !isfinite(depth) || !isfinite(depthD)) {
                change[index] = 0;
                change[index + (size_t)W] = 0;
            }
        }
    /* 2. distance map */
    float *dist = (float *)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; ++i) dist[i] = change[i] == 0 ? 0.0f : (float)(W + H);
    for (int ri = 1; ri < H; ++ri) {
        const float *previous_row = dist + (size_t)(ri - 1) * (size_t)W;
        float *current_row = dist + (size_t)ri * (size_t)W;
        for (int ci = 1; ci < W; ++ci) {
            /* previous_row[W] at ci = W - 1 is the first pixel of the current row (rows are contiguous) */
            const float upLeft = previous_row[ci - 1] + 1.4f, up = previous_row[ci] + 1.0f, upRight = previous_row[ci + 1] + 1.4f;
            const float left = current_row[ci - 1] + 1.0f, center = current_row[ci];
            const float a = upLeft < up ? upLeft : up, b2 = left < upRight ? left : upRight;   /* std::min(std::min(upLeft, up), std::min(left, upRight)) */
            const float minValue = a < b2 ? a : b2;
            if (minValue < center) current_row[ci] = minValue;
        }
    }
    for (int ri = H - 2; ri >= 0; --ri) {
        const float *next_row = dist + (size_t)(ri + 1) * (size_t)W;
        float *current_row = dist + (size_t)ri * (size_t)W;
        for (int ci = W - 2; ci >= 0; --ci) {
            /* next_row[-1] at ci = 0 is the last pixel of the current row (rows are contiguous) */
            const float lowerLeft = next_row[ci - 1] + 1.4f, lower = next_row[ci] + 1.0f, lowerRight = next_row[ci + 1] + 1.4f;
            const float right = current_row[ci + 1] + 1.0f, center = current_row[ci];
            const float a = lowerLeft < lower ? lowerLeft : lower, b2 = right < lowerRight ? right : lowerRight;
            const float minValue = a < b2 ? a : b2;
            if (minValue < center) current_row[ci] = minValue;
        }
    }
    /* 3. integral image (width + 1) x (height + 1), 3 doubles per element */
    double *ii = (double *)calloc((size_t)(W + 1) * (size_t)(H + 1) * 3, sizeof(double));
    for (int r = 0; r < H; ++r) {
        const double *prev = ii + 3 * (size_t)r * (size_t)(W + 1);
        double *cur = ii + 3 * (size_t)(r + 1) * (size_t)(W + 1);
        for (int c = 0; c < W; ++c) {
            const float *e = xyz + 3 * ((size_t)r * (size_t)W + (size_t)c);
            for (int a = 0; a < 3; ++a) cur[3 * (c + 1) + a] = (prev[3 * (c + 1) + a] + cur[3 * c + a]) - prev[3 * c + a];
            const float s = e[0] + (e[1] + e[2]);     /* Eigen's fixed-size Vector3f::sum() reduces as x + (y + z) */
            if (isfinite(s))
                for (int a = 0; a < 3; ++a) cur[3 * (c + 1) + a] += (double)e[a];
        }
    }
    /* 4. normals of the pixels inside the border */
    for (int ri = border; ri < H - border; ++ri)
        for (int ci = border; ci < W - border; ++ci) {
            const size_t index = (size_t)ri * (size_t)W + (size_t)ci;
            const float *pt = xyz + 3 * index;
            if (!isfinite(pt[2])) continue;
            const float smoothing = dist[index] < smoothing_size ? dist[index] : smoothing_size;
            if (!(smoothing > 2.0f)) continue;
            const int rw = (int)smoothing, rh = (int)smoothing, rw2 = rw / 2, rh2 = rh / 2;
            double s1[3], s0[3], gx[3], gy[3];
            ii_sum(ii, W, ci + rw2, ri - rh2, 1, rh, s1);
            ii_sum(ii, W, ci - rw2, ri - rh2, 1, rh, s0);
            for (int a = 0; a < 3; ++a) gx[a] = s1[a] - s0[a];
            ii_sum(ii, W, ci - rw2, ri + rh2, rw, 1, s1);
            ii_sum(ii, W, ci - rw2, ri - rh2, rw, 1, s0);
            for (int a = 0; a < 3; ++a) gy[a] = s1[a] - s0[a];
            double nv[3] = {gy[1] * gx[2] - gy[2] * gx[1], gy[2] * gx[0] - gy[0] * gx[2], gy[0] * gx[1] - gy[1] * gx[0]};
            const double len = (nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2];
            if (len == 0.0) continue;
            const double root = sqrt(len);
            float nx = (float)(nv[0] / root), ny = (float)(nv[1] / root), nz = (float)(nv[2] / root);
            /* pcl::flipNormalTowardsViewpoint (float overload) */
            const float vx = vp[0] - pt[0], vy = vp[1] - pt[1], vz = vp[2] - pt[2];
            const float cos_theta = (vx * nx + vy * ny + vz * nz);
            if (cos_theta < 0) { nx *= -1; ny *= -1; nz *= -1; }
            normals_out[3 * index] = nx; normals_out[3 * index + 1] = ny; normals_out[3 * index + 2] = nz;
        }
    free(ii);
    free(dist);
    free(change);
}
