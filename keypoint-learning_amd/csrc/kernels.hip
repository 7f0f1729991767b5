// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of libkpl.
//
// Built with -ffp-contract=off: the reference arithmetic (and the oracle's) is separate float
// mul / add / div / sqrt, so no FMA contraction anywhere in this file; hipcc's default
// correctly rounded fp32 division and square root are kept (no -ffast-math).
//
// Path: /root/reference/include/impl/KeypointLearning.hpp:179-376 and
// /root/reference/src/KeypointLearning.cpp:41-92; per-kernel citations below.
#include "kernels.h"

#include <cmath>

namespace kpl {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ bool finite3(float x, float y, float z) {
    return isfinite(x) && isfinite(y) && isfinite(z);
}

// Normative cell function (DESIGN.md "canonical order"): float subtract, IEEE float divide,
// floor, clamp.  Monotonic in v, which is what makes the box search below exact.
__device__ __forceinline__ int cell_coord(float v, float mn, float h, int dim) {
    float t = floorf((v - mn) / h);
    if (!(t >= 0.0f)) return 0;
    if (t >= (float)dim) return dim - 1;
    return (int)t;
}

__device__ __forceinline__ uint32_t enc_f32(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

inline float dec_f32(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    union { uint32_t u; float f; } cvt;
    cvt.u = u;
    return cvt.f;
}

__device__ __forceinline__ const float *point_at(const char *base, size_t stride, int i) {
    return reinterpret_cast<const float *>(base + (size_t)i * stride);
}

// ---------------------------------------------------------------------------------------------
// bounding box of the finite points
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_kernel(const char *xyz, size_t stride, int n,
                                                   uint32_t *bbox) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *p = point_at(xyz, stride, i);
        float x = p[0], y = p[1], z = p[2];
        if (finite3(x, y, z)) {
            mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
            mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
            mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        for (int off = kWave / 2; off > 0; off >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], off));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off));
        }
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (mn[k] <= mx[k]) {
                atomicMin(&bbox[k], enc_f32(mn[k]));
                atomicMax(&bbox[3 + k], enc_f32(mx[k]));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cell id per point + population count per cell
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cell_count_kernel(const char *xyz, size_t stride, int n,
                                                         GridDesc g, int *cid, int *cnt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = point_at(xyz, stride, i);
    float x = p[0], y = p[1], z = p[2];
    int c = -1;
    if (finite3(x, y, z)) {
        int cx = cell_coord(x, g.mn[0], g.h, g.dims[0]);
        int cy = cell_coord(y, g.mn[1], g.h, g.dims[1]);
        int cz = cell_coord(z, g.mn[2], g.h, g.dims[2]);
        c = (cz * g.dims[1] + cy) * g.dims[0] + cx;
        atomicAdd(&cnt[c], 1);
    }
    cid[i] = c;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan (3 launches): chunk sums -> scan of sums -> per-chunk scan with carry
// ---------------------------------------------------------------------------------------------
constexpr int kScanBlock = 256;
constexpr int kScanPerThread = 16;
constexpr int kScanChunk = kScanBlock * kScanPerThread;

__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
    __shared__ int wave_sum[kScanBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    int incl = v;
    for (int off = 1; off < kWave; off <<= 1) {
        int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) wave_sum[wid] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < kScanBlock / kWave; ++w) {
        int s = wave_sum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_sums_kernel(const int *in, int len, int *sums) {
    const int base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < len) s += in[base + k];
    int tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanBlock) void scan_top_kernel(int *sums, int nb) {
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += kScanBlock) {
        int i = b0 + threadIdx.x;
        int v = i < nb ? sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot);
        if (i < nb) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}

__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const int *in, int *out, int len,
                                                                const int *sums, int nb) {
    const int base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int v[kScanPerThread];
    int s = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        v[k] = base + k < len ? in[base + k] : 0;
        s += v[k];
    }
    int tot;
    int run = block_exclusive_scan(s, &tot) + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        if (base + k < len) out[base + k] = run;
        run += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[len] = sums[nb];
}

// ---------------------------------------------------------------------------------------------
// counting sort, made deterministic: scatter in arrival order, then rank inside the cell by
// original index (ascending), and store the point + normal at its canonical position
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_kernel(const int *cid, int n, int *cursor,
                                                      int *tmp_idx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = cid[i];
    if (c < 0) return;
    int slot = atomicAdd(&cursor[c], 1);
    tmp_idx[slot] = i;
}

__global__ __launch_bounds__(256) void rank_store_kernel(const char *xyz, size_t xs,
                                                         const char *nrm, size_t ns, int n,
                                                         GridDesc g, const int *cid,
                                                         const int *cell_start, const int *tmp_idx,
                                                         float4 *pts, float4 *nrmo, int *pos_of) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n && cid[s] < 0) pos_of[s] = -1;   // non-finite original point s
    const int nfinite = cell_start[g.ncells];
    if (s >= nfinite) return;
    const int i = tmp_idx[s];
    const int c = cid[i];
    const int s0 = cell_start[c], s1 = cell_start[c + 1];
    int rank = 0;
    for (int t = s0; t < s1; ++t) rank += (tmp_idx[t] < i);
    const int pos = s0 + rank;
    const float *p = point_at(xyz, xs, i);
    const float *q = point_at(nrm, ns, i);
    pts[pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
    float nx = q[0], ny = q[1], nz = q[2];
    nrmo[pos] = make_float4(nx, ny, nz, finite3(nx, ny, nz) ? 1.0f : 0.0f);
    pos_of[i] = pos;
}

// ---------------------------------------------------------------------------------------------
// Soft assignment, /root/reference/src/KeypointLearning.cpp:41-65 and :68-92.  dim and dim/2
// are per-launch constants computed on the host with the same float operations.  The
// reference's assert on the index range is replaced by a clamp (no effect on in-range values).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void soft_pair(int n, float v, float dim, float half_dim, int &k,
                                          int &p, float &w) {
    k = (int)floorf(v / dim);
    if (k == n) k--;
    k = min(max(k, 0), n - 1);
    float center = ((float)k * dim) + half_dim;
    float wt = v - center;
    wt = wt / dim;
    p = (wt > 0) ? k + 1 : k - 1;
    if (p == -1) p = 0;
    if (p == n) p = k;
    w = fabsf(wt);
}

struct CellBox {
    int lo[3], hi[3];
};

__device__ __forceinline__ CellBox make_box(const GridDesc &g, float x, float y, float z,
                                            float rr) {
    CellBox b;
    b.lo[0] = cell_coord(x - rr, g.mn[0], g.h, g.dims[0]);
    b.hi[0] = cell_coord(x + rr, g.mn[0], g.h, g.dims[0]);
    b.lo[1] = cell_coord(y - rr, g.mn[1], g.h, g.dims[1]);
    b.hi[1] = cell_coord(y + rr, g.mn[1], g.h, g.dims[1]);
    b.lo[2] = cell_coord(z - rr, g.mn[2], g.h, g.dims[2]);
    b.hi[2] = cell_coord(z + rr, g.mn[2], g.h, g.dims[2]);
    return b;
}

// FLANN L2_Simple<float>: d = dx*dx; d += dy*dy; d += dz*dz
__device__ __forceinline__ float dist2(float px, float py, float pz, const float4 &q) {
    float dx = px - q.x, dy = py - q.y, dz = pz - q.z;
    float d = dx * dx;
    d += dy * dy;
    d += dz * dz;
    return d;
}

// computePointFeatures, hpp:321-376.  One lane = one query point; the A x B histogram of the
// lane lives in LDS as H[c * BLOCK + tid] (bank = tid mod 32: conflict free).  Neighbors are
// visited in canonical order (rows of cells ascending, storage positions ascending), the first
// accepted one is dropped (hpp:336 starts at neigh_indx = 1).  Returns K_f.
template <int BLOCK>
__device__ __forceinline__ int point_features(const float4 *__restrict__ pts,
                                              const float4 *__restrict__ nrm,
                                              const int *__restrict__ cell_start,
                                              const GridDesc &g, const FeatDesc &f, float4 p,
                                              float4 np, float *H) {
    const int tid = threadIdx.x;
    for (int c = 0; c < f.F; ++c) H[c * BLOCK + tid] = 0.0f;                     // hpp:325
    const CellBox b = make_box(g, p.x, p.y, p.z, f.rr);
    int seen = 0;
    for (int cz = b.lo[2]; cz <= b.hi[2]; ++cz) {
        for (int cy = b.lo[1]; cy <= b.hi[1]; ++cy) {
            const int row = (cz * g.dims[1] + cy) * g.dims[0];
            const int t0 = cell_start[row + b.lo[0]];
            const int t1 = cell_start[row + b.hi[0] + 1];
            for (int t = t0; t < t1; ++t) {
                const float4 q = pts[t];
                const float d2 = dist2(p.x, p.y, p.z, q);
                if (!(d2 < f.r2)) continue;                                        // strict
                if (seen++ == 0) continue;                                         // hpp:336
                const float4 nq = nrm[t];
                if (nq.w == 0.0f) continue;                                        // hpp:338
                const float dot = np.x * nq.x + (np.y * nq.y + np.z * nq.z);       // hpp:342
                float cosine = 1 - dot;
                int a, ap, bi, bp;
                float aw, bw;
                soft_pair(f.A, sqrtf(d2), f.ann_dim, f.ann_half, a, ap, aw);       // hpp:345
                if (cosine < 0) cosine = 0;                                        // cpp:70-73
                if (cosine > 2) cosine = 2;
                soft_pair(f.B, cosine, f.bin_dim, f.bin_half, bi, bp, bw);         // hpp:348
                const float w00 = (1 - bw) * (1 - aw);
                const float w01 = bw * (1 - aw);
                const float w10 = (1 - bw) * aw;
                const float w11 = bw * aw;
                float *h0 = H + (a * f.B) * BLOCK + tid;
                float *h1 = H + (ap * f.B) * BLOCK + tid;
                h0[bi * BLOCK] += w00;                                             // hpp:350
                h0[bp * BLOCK] += w01;                                             // hpp:351
                h1[bi * BLOCK] += w10;                                             // hpp:354
                h1[bp * BLOCK] += w11;                                             // hpp:355
            }
        }
    }
    for (int a = 0; a < f.A; ++a) {                                                // hpp:360-370
        float *h = H + (a * f.B) * BLOCK + tid;
        float s = 0.0f;
        for (int k = 0; k < f.B; ++k) {
            float v = h[k * BLOCK];
            s += v * v;
        }
        const float nr = sqrtf(s);
        if (nr > 0)
            for (int k = 0; k < f.B; ++k) h[k * BLOCK] = h[k * BLOCK] / nr;
    }
    return seen;
}

// runForest, hpp:267-296 + cv::ml::RTrees::predict(PREDICT_SUM) restated (hpp:281): per tree
// walk "val <= thr ? left : right", double sum of leaf values, (float)sum,
// score = 1 - sum / (T * 1.0f).
template <int BLOCK, bool STATS>
__global__ __launch_bounds__(BLOCK) void score_kernel(const float4 *__restrict__ pts,
                                                      const float4 *__restrict__ nrm,
                                                      const int *__restrict__ cell_start,
                                                      GridDesc g, FeatDesc f, ForestDev forest,
                                                      float *__restrict__ score_sorted,
                                                      float *__restrict__ scores,
                                                      StatsDev *stats) {
    extern __shared__ float H[];
    const int s = blockIdx.x * BLOCK + threadIdx.x;
    const int nfinite = cell_start[g.ncells];
    if (s >= nfinite) return;
    const float4 p = pts[s];
    const float4 np = nrm[s];
    float score = NAN;
    if (np.w != 0.0f) {                                                            // hpp:277
        const int kf = point_features<BLOCK>(pts, nrm, cell_start, g, f, p, np, H);
        double sum = 0.0;
        int depth = 0;
        for (int t = 0; t < forest.ntrees; ++t) {
            uint32_t nd = forest.roots[t];
            for (;;) {
                const uint2 node = forest.nodes[nd];
                const uint32_t var = node.y >> 24;
                if (STATS) ++depth;
                if (var == 255u) {
                    sum += (double)__uint_as_float(node.x);
                    break;
                }
                const float val = H[var * BLOCK + threadIdx.x];
                nd = (node.y & 0x00ffffffu) + (val <= __uint_as_float(node.x) ? 0u : 1u);
            }
        }
        const float fsum = (float)sum;
        score = 1 - (fsum / (forest.ntrees * 1.0f));                               // hpp:287
        if (STATS) {
            atomicAdd(&stats->sum_kf, (unsigned long long)kf);
            atomicAdd(&stats->sum_depth, (unsigned long long)depth);
            atomicAdd(&stats->n_scored, 1ull);
        }
    }
    score_sorted[s] = score;
    if (scores) scores[__float_as_int(p.w)] = score;
}

// computePointsForTrainingFeatures, hpp:299-318: same feature code, sparse query list.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void features_kernel(const float4 *__restrict__ pts,
                                                         const float4 *__restrict__ nrm,
                                                         const int *__restrict__ cell_start,
                                                         const int *__restrict__ pos_of,
                                                         GridDesc g, FeatDesc f,
                                                         const int *__restrict__ query, int m,
                                                         int n, float *__restrict__ out) {
    extern __shared__ float H[];
    const int qi = blockIdx.x * BLOCK + threadIdx.x;
    if (qi >= m) return;
    const int i = query[qi];
    const int s = (i >= 0 && i < n) ? pos_of[i] : -1;
    float *o = out + (size_t)qi * f.F;
    if (s < 0) {
        for (int c = 0; c < f.F; ++c) o[c] = NAN;
        return;
    }
    point_features<BLOCK>(pts, nrm, cell_start, g, f, pts[s], nrm[s], H);
    for (int c = 0; c < f.F; ++c) o[c] = H[c * BLOCK + threadIdx.x];
}

__global__ void fill_f32_kernel(float *p, float v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// detectKeypoints, hpp:197-256 with draws_remove == false (order-independent predicate):
// keypoint <=> score >= thr (float promoted to double, hpp:207) and no neighbor within r_nms
// has a strictly greater score (hpp:219).  non_maxima == 0: every scoreable point (hpp:189-196).
template <bool STATS>
__global__ __launch_bounds__(256) void nms_kernel(const float4 *__restrict__ pts,
                                                  const int *__restrict__ cell_start, GridDesc g,
                                                  NmsDesc nd, const float *__restrict__ score_sorted,
                                                  int *__restrict__ flags, StatsDev *stats) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int nfinite = cell_start[g.ncells];
    if (s >= nfinite) return;
    const float si = score_sorted[s];
    if (!isfinite(si)) return;                                                     // hpp:206
    const float4 p = pts[s];
    const int orig = __float_as_int(p.w);
    if (!nd.non_maxima) {
        flags[orig] = 1;
        return;
    }
    if ((double)si < nd.thr) return;                                               // hpp:207
    const CellBox b = make_box(g, p.x, p.y, p.z, nd.rr);
    bool is_max = true;
    int kn = 0;
    for (int cz = b.lo[2]; cz <= b.hi[2] && (is_max || STATS); ++cz) {
        for (int cy = b.lo[1]; cy <= b.hi[1] && (is_max || STATS); ++cy) {
            const int row = (cz * g.dims[1] + cy) * g.dims[0];
            const int t0 = cell_start[row + b.lo[0]];
            const int t1 = cell_start[row + b.hi[0] + 1];
            for (int t = t0; t < t1; ++t) {
                if (dist2(p.x, p.y, p.z, pts[t]) < nd.r2) {
                    if (STATS) ++kn;
                    if (si < score_sorted[t]) {                                    // hpp:219
                        is_max = false;
                        if (!STATS) break;
                    }
                }
            }
        }
    }
    if (STATS) {
        atomicAdd(&stats->sum_kn, (unsigned long long)kn);
        atomicAdd(&stats->n_thresholded, 1ull);
    }
    if (is_max) flags[orig] = 1;                                                   // hpp:252-253
}

__global__ __launch_bounds__(256) void compact_kernel(const int *flags, const int *prefix, int n,
                                                      int *kp_idx, int kp_cap, int *kp_count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *kp_count = prefix[n];
    if (i >= n) return;
    if (flags[i]) {
        int pos = prefix[i];
        if (pos < kp_cap) kp_idx[pos] = i;
    }
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// =============================================================================================
// launch wrappers
// =============================================================================================
void launch_bbox(const char *xyz, size_t stride, int n, uint32_t *bbox, hipStream_t st) {
    static const uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    (void)hipMemcpyAsync(bbox, init, sizeof(init), hipMemcpyHostToDevice, st);
    if (n <= 0) return;
    int blocks = div_up(n, 256);
    if (blocks > 2048) blocks = 2048;
    bbox_kernel<<<blocks, 256, 0, st>>>(xyz, stride, n, bbox);
}

void decode_bbox(const uint32_t *enc, float *mn, float *mx) {
    for (int k = 0; k < 3; ++k) {
        mn[k] = dec_f32(enc[k]);
        mx[k] = dec_f32(enc[3 + k]);
    }
}

void launch_cell_count(const char *xyz, size_t stride, int n, GridDesc g, int *cid, int *cnt,
                       hipStream_t st) {
    (void)hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(g.ncells + 1), st);
    if (n <= 0) return;
    cell_count_kernel<<<div_up(n, 256), 256, 0, st>>>(xyz, stride, n, g, cid, cnt);
}

void launch_exclusive_scan(const int *in, int *out, int len, int *tmp, hipStream_t st) {
    const int nb = len > 0 ? div_up(len, kScanChunk) : 1;
    scan_sums_kernel<<<nb, kScanBlock, 0, st>>>(in, len, tmp);
    scan_top_kernel<<<1, kScanBlock, 0, st>>>(tmp, nb);
    scan_apply_kernel<<<nb, kScanBlock, 0, st>>>(in, out, len, tmp, nb);
}

void launch_scatter(const int *cid, int n, const int *cell_start, int *cursor, int *tmp_idx,
                    hipStream_t st) {
    (void)cell_start;
    if (n <= 0) return;
    scatter_kernel<<<div_up(n, 256), 256, 0, st>>>(cid, n, cursor, tmp_idx);
}

void launch_rank_store(const char *xyz, size_t xs, const char *nrm, size_t ns, int n, GridDesc g,
                       const int *cid, const int *cell_start, const int *tmp_idx, float4 *pts,
                       float4 *nrmo, int *pos_of, hipStream_t st) {
    if (n <= 0) return;
    rank_store_kernel<<<div_up(n, 256), 256, 0, st>>>(xyz, xs, nrm, ns, n, g, cid, cell_start,
                                                      tmp_idx, pts, nrmo, pos_of);
}

int score_block_size(int F) {
    if (F <= 32) return 256;
    if (F <= 64) return 128;
    return 64;
}

template <int BLOCK>
static void launch_score_b(const float4 *pts, const float4 *nrm, const int *cell_start,
                           GridDesc g, FeatDesc f, ForestDev forest, int n, float *score_sorted,
                           float *scores, StatsDev *stats, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)f.F * BLOCK;
    if (stats) {
        (void)hipFuncSetAttribute((const void *)score_kernel<BLOCK, true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        score_kernel<BLOCK, true><<<div_up(n, BLOCK), BLOCK, lds, st>>>(
            pts, nrm, cell_start, g, f, forest, score_sorted, scores, stats);
    } else {
        (void)hipFuncSetAttribute((const void *)score_kernel<BLOCK, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        score_kernel<BLOCK, false><<<div_up(n, BLOCK), BLOCK, lds, st>>>(
            pts, nrm, cell_start, g, f, forest, score_sorted, scores, stats);
    }
}

void launch_score(const float4 *pts, const float4 *nrm, const int *cell_start, GridDesc g,
                  FeatDesc f, ForestDev forest, int n, float *score_sorted, float *scores,
                  StatsDev *stats, hipStream_t st) {
    if (n <= 0) return;
    switch (score_block_size(f.F)) {
        case 256: launch_score_b<256>(pts, nrm, cell_start, g, f, forest, n, score_sorted, scores, stats, st); break;
        case 128: launch_score_b<128>(pts, nrm, cell_start, g, f, forest, n, score_sorted, scores, stats, st); break;
        default:  launch_score_b<64>(pts, nrm, cell_start, g, f, forest, n, score_sorted, scores, stats, st); break;
    }
}

template <int BLOCK>
static void launch_features_b(const float4 *pts, const float4 *nrm, const int *cell_start,
                              const int *pos_of, GridDesc g, FeatDesc f, const int *query, int m,
                              int n, float *out, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)f.F * BLOCK;
    (void)hipFuncSetAttribute((const void *)features_kernel<BLOCK>,
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    features_kernel<BLOCK><<<div_up(m, BLOCK), BLOCK, lds, st>>>(pts, nrm, cell_start, pos_of, g,
                                                                  f, query, m, n, out);
}

void launch_features(const float4 *pts, const float4 *nrm, const int *cell_start,
                     const int *pos_of, GridDesc g, FeatDesc f, const int *query, int m, int n,
                     float *out, hipStream_t st) {
    if (m <= 0) return;
    switch (score_block_size(f.F)) {
        case 256: launch_features_b<256>(pts, nrm, cell_start, pos_of, g, f, query, m, n, out, st); break;
        case 128: launch_features_b<128>(pts, nrm, cell_start, pos_of, g, f, query, m, n, out, st); break;
        default:  launch_features_b<64>(pts, nrm, cell_start, pos_of, g, f, query, m, n, out, st); break;
    }
}

void launch_fill_f32(float *p, float v, int n, hipStream_t st) {
    if (n <= 0) return;
    fill_f32_kernel<<<div_up(n, 256), 256, 0, st>>>(p, v, n);
}

void launch_nms(const float4 *pts, const int *cell_start, GridDesc g, NmsDesc nd,
                const float *score_sorted, int n, int *flags, StatsDev *stats, hipStream_t st) {
    (void)hipMemsetAsync(flags, 0, sizeof(int) * (size_t)(n + 1), st);
    if (n <= 0) return;
    if (stats)
        nms_kernel<true><<<div_up(n, 256), 256, 0, st>>>(pts, cell_start, g, nd, score_sorted, flags, stats);
    else
        nms_kernel<false><<<div_up(n, 256), 256, 0, st>>>(pts, cell_start, g, nd, score_sorted, flags, stats);
}

void launch_compact(const int *flags, const int *prefix, int n, int *kp_idx, int kp_cap,
                    int *kp_count, hipStream_t st) {
    compact_kernel<<<div_up(n > 0 ? n : 1, 256), 256, 0, st>>>(flags, prefix, n, kp_idx, kp_cap, kp_count);
}

}  // namespace kpl
