// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of libkpl.
//
// Built with -ffp-contract=off: the reference arithmetic (and the oracle's) is separate float
// mul / add / div / sqrt, so no FMA contraction anywhere in this file; hipcc's default
// correctly rounded fp32 division and square root are kept (no -ffast-math).
//
// Path: /root/reference/include/impl/KeypointLearning.hpp:179-376 and
// /root/reference/src/KeypointLearning.cpp:41-92; per-kernel citations below.
#include "kernels.h"
#include "exact_math.h"
#include "soft_pair.h"

#ifdef KPL_ABLATE
#error "KPL_ABLATE timing experiments are not part of libkpl: build them from a scratch copy of this file"
#endif

#include <cmath>
#include <cstdlib>
#include <cstring>

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

__device__ __forceinline__ const float *point_at(const char *base, size_t stride, int i) {
    return reinterpret_cast<const float *>(base + (size_t)i * stride);
}

// ---------------------------------------------------------------------------------------------
// bounding box of the finite points
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_kernel(Batch b) {
    const ViewDev &v = b.view[blockIdx.y];
    const char *xyz = v.xyz;
    const size_t stride = v.xs;
    const int n = v.n;
    uint32_t *bbox = v.ds->bbox;
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
    // one atomic per block and component: tens of thousands of same-address atomics serialise
    __shared__ float red[2][3][256 / kWave];
    const int wid = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            red[0][k][wid] = mn[k];
            red[1][k][wid] = mx[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        float a = red[0][k][0], b = red[1][k][0];
        for (int w = 1; w < 256 / kWave; ++w) {
            a = fminf(a, red[0][k][w]);
            b = fmaxf(b, red[1][k][w]);
        }
        if (a <= b) {
            atomicMin(&bbox[k], enc_f32(a));
            atomicMax(&bbox[3 + k], enc_f32(b));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// grid descriptor from the bounding box, on the device (the host never waits for it).  Same float
// operations as the CPU restatement: dims[k] = (int)floorf((max - min) / h) + 1.
// Also re-arms the bounding-box accumulators for the next call.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dec_f32_dev(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

__global__ void grid_setup_kernel(Batch b) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const ViewDev &v = b.view[blockIdx.y];
    DevState *ds = v.ds;
    float h = v.cell;
    const int n = v.n, cells_cap = v.cells_cap;
    GridDesc g;
    const bool any = n > 0 && ds->bbox[0] != 0xffffffffu;
    if (!(h > 0.0f)) {
        // no radius given (cloud resolution): about two points per cell on a surface-like cloud
        float e[3];
        for (int k = 0; k < 3; ++k) e[k] = any ? dec_f32_dev(ds->bbox[3 + k]) - dec_f32_dev(ds->bbox[k]) : 0.0f;
        if (e[0] < e[1]) { float t = e[0]; e[0] = e[1]; e[1] = t; }
        if (e[1] < e[2]) { float t = e[1]; e[1] = e[2]; e[2] = t; }
        if (e[0] < e[1]) { float t = e[0]; e[0] = e[1]; e[1] = t; }
        h = sqrtf(e[0] * (e[1] > 0.0f ? e[1] : e[0]) / (float)(n > 0 ? n : 1) * 2.0f);
        if (!(h > 0.0f)) h = 1.0f;
    }
    g.h = h;
    long long nc = any ? 1 : 0;
    int status = 0;
    for (int k = 0; k < 3; ++k) {
        float mn = any ? dec_f32_dev(ds->bbox[k]) : 0.0f;
        const float mx = any ? dec_f32_dev(ds->bbox[3 + k]) : 0.0f;
        if (v.has_origin) {             // the caller's grid frame (a slab of a larger cloud)
            if (any && !(v.origin[k] <= mn)) status = kStatusBadOrigin;
            mn = v.origin[k];
        }
        g.mn[k] = mn;
        g.dims[k] = 0;
        if (!any) continue;
        if (status != 0) {
            nc = 0;
            break;
        }
        const float t = floorf((mx - mn) / h);
        if (!(t < 1.0e9f)) {
            status = kStatusGridTooLarge;
            nc = 0;
            break;
        }
        g.dims[k] = (int)t + 1;
        nc *= g.dims[k];
        if (nc > kMaxGridCells) {
            status = kStatusGridTooLarge;
            nc = 0;
            break;
        }
    }
    ds->ncells_needed = (int)nc;
    if (status == 0 && nc > cells_cap) status = kStatusCellCapacity;
    if (status != 0) {
        nc = 0;
        g.dims[0] = g.dims[1] = g.dims[2] = 0;
    }
    g.ncells = (int)nc;
    ds->grid = g;
    ds->status = status;
    // buckets of the index sort: 2^bshift consecutive cells each, at most kBuckets of them
    int bshift = 0;
    while (nc > 0 && ((nc - 1) >> bshift) >= kBuckets) ++bshift;
    ds->bshift = bshift;
    ds->nbuckets = nc > 0 ? (int)((nc - 1) >> bshift) + 1 : 0;
    for (int k = 0; k < 3; ++k) {
        ds->bbox[k] = 0xffffffffu;
        ds->bbox[3 + k] = 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// exclusive scan (3 launches): chunk sums -> scan of sums -> per-chunk scan with carry
// ---------------------------------------------------------------------------------------------
// One wave per workgroup: in the pipelined mode (two batches in flight) these kernels queue behind the other batch's feature
// kernel, which fills the register files; a 256-thread workgroup needs four free wave slots on ONE CU at once, a 64-thread one
// fits any slot that comes free.  compact_scan_kernel under the bench: avg 39.6 -> 20.5 us, max 520 -> 50 (profiles/r06_notes.md).
// (compact_scan_kernel<64>, the NMS kernel and, for a batch, bucket_total / bucket_offsets; the three-kernel scan of the draws
// pass keeps 256.)
constexpr int kScanBlock = 256;
constexpr int kCompactBatchBlock = 64;        // threads of compact_scan_kernel for a batch; kScanBlock for one view alone (fewer, longer blocks: the shorter look-back chain)
constexpr int kScanPerThread = 16;
constexpr int kScanChunk = kScanBlock * kScanPerThread;

template <int BLOCK = kScanBlock>
__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
    __shared__ int wave_sum[BLOCK / kWave];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    int incl = v;
    for (int off = 1; off < kWave; off <<= 1) {
        int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) wave_sum[wid] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < BLOCK / kWave; ++w) {
        int s = wave_sum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// `len` is the launch-time upper bound; when dlen is given the actual length is min(*dlen, len)
__device__ __forceinline__ int scan_len(const int *dlen, int len) {
    return dlen ? min(*dlen, len) : len;
}

// one exclusive scan per view of a batch: in[0..L) -> out[0..L], out[L] = total (also into out2 if
// given), L = min(*dlen, len) when dlen is given (len = launch-time upper bound); tmp holds
// >= len/4096 + 2 ints
struct ScanJob {
    int *in, *out, *out2;
    const int *dlen;
    int len;
    int *tmp;
};
struct ScanJobs {
    ScanJob job[kMaxBatch];
    int zero_in;   // clear the input behind the read
    int match;     // < 0: scan the values, >= 0: scan the predicate (value == match)
};

// match < 0: scan the values themselves; match >= 0: scan the predicate (value == match)
__device__ __forceinline__ int scan_value(int v, int match) { return match < 0 ? v : (v == match ? 1 : 0); }

__global__ __launch_bounds__(kScanBlock) void scan_sums_kernel(ScanJobs jobs) {
    const ScanJob &job = jobs.job[blockIdx.y];
    const int *in = job.in;
    int *sums = job.tmp;
    const int match = jobs.match;
    const int len = scan_len(job.dlen, job.len);
    if ((int)blockIdx.x * kScanChunk >= job.len && blockIdx.x > 0) return;   // beyond this view's launch bound
    const int base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int s = 0;
    if (base + kScanPerThread <= len) {            // the thread's 16 values as four 16-byte loads (64-byte aligned)
        const int4 *q = reinterpret_cast<const int4 *>(in + base);
#pragma unroll
        for (int k = 0; k < kScanPerThread / 4; ++k) {
            const int4 x = q[k];
            s += scan_value(x.x, match) + scan_value(x.y, match) + scan_value(x.z, match) + scan_value(x.w, match);
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k)
            if (base + k < len) s += scan_value(in[base + k], match);
    }
    int tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanBlock) void scan_top_kernel(ScanJobs jobs) {
    const ScanJob &job = jobs.job[blockIdx.y];
    int *sums = job.tmp;
    const int nb = job.len > 0 ? (job.len + kScanChunk - 1) / kScanChunk : 1;
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

// out[i] = exclusive prefix, out[len] = total; the same goes to out2 when given; the input is
// zeroed behind the read when zero_in is set (self-cleaning counters: no memset per call)
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(ScanJobs jobs) {
    const ScanJob &job = jobs.job[blockIdx.y];
    int *in = job.in, *out = job.out, *out2 = job.out2;
    const int *sums = job.tmp;
    const int nb = job.len > 0 ? (job.len + kScanChunk - 1) / kScanChunk : 1;
    const int zero_in = jobs.zero_in, match = jobs.match;
    if ((int)blockIdx.x >= nb) return;
    const int len = scan_len(job.dlen, job.len);
    const int base = blockIdx.x * kScanChunk + threadIdx.x * kScanPerThread;
    int v[kScanPerThread];
    int s = 0;
    const bool whole = base + kScanPerThread <= len;      // 16 values = four 16-byte loads / stores (64-byte aligned)
    if (whole) {
        int4 *q = reinterpret_cast<int4 *>(in + base);
#pragma unroll
        for (int k = 0; k < kScanPerThread / 4; ++k) {
            const int4 x = q[k];
            v[4 * k] = scan_value(x.x, match);
            v[4 * k + 1] = scan_value(x.y, match);
            v[4 * k + 2] = scan_value(x.z, match);
            v[4 * k + 3] = scan_value(x.w, match);
            if (zero_in) q[k] = make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) s += v[k];
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) {
            v[k] = base + k < len ? scan_value(in[base + k], match) : 0;
            if (zero_in && base + k < len) in[base + k] = 0;
            s += v[k];
        }
    }
    int tot;
    int run = block_exclusive_scan(s, &tot) + sums[blockIdx.x];
    if (whole) {
        int4 *o = reinterpret_cast<int4 *>(out + base), *o2 = out2 ? reinterpret_cast<int4 *>(out2 + base) : nullptr;
#pragma unroll
        for (int k = 0; k < kScanPerThread / 4; ++k) {
            int4 r;
            r.x = run;
            r.y = r.x + v[4 * k];
            r.z = r.y + v[4 * k + 1];
            r.w = r.z + v[4 * k + 2];
            run = r.w + v[4 * k + 3];
            o[k] = r;
            if (o2) o2[k] = r;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) {
            if (base + k < len) {
                out[base + k] = run;
                if (out2) out2[base + k] = run;
            }
            run += v[k];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[len] = sums[nb];
        if (out2) out2[len] = sums[nb];
    }
}

// ---------------------------------------------------------------------------------------------
// Index build ("initCompute"): a STABLE two-level counting sort of the points by cell id, so that
// the storage order is the canonical one (cell id ascending, original index ascending inside a
// cell) by construction -- no global atomics, no ranking pass.
//   level 1  the cell id space is cut into nbuckets <= 1024 buckets of 2^bshift consecutive cells
//            (DevState, computed on the device with the grid).  Chunks of kSortChunk consecutive
//            points (one wave each) count their points per bucket (bucket_hist_kernel), the
//            (chunk, bucket) table is turned into offsets column by column (bucket_total_kernel,
//            bucket_offsets_kernel), and the chunks move their (index, cell) pairs to the bucket's range, in index order
//            (bucket_scatter_kernel).  Points that are not finite go to an extra bin at the end.
//   level 2  one wave per bucket (cell_sort_store_kernel): per-cell counts of the bucket in LDS,
//            scan -> cell_start[], then the bucket's pairs once more in index order: storage
//            position = start of the cell + points of that cell seen so far; the point and its
//            normal are gathered and stored there, pos_of[index] = position.
// "In index order" inside a wave: the 64 pairs of a round are ranked among the lanes with the same
// key by lane number (match_rank) and the running count of the key is advanced once per round.
// ---------------------------------------------------------------------------------------------
constexpr int kSortChunk = 1024;         // points per chunk of level 1 (16 rounds of one wave) ...
constexpr int kSortChunkSmall = 512;     // ... or half of that when the whole launch is small (sort_chunk_points): twice the waves for
                                         // kernels that are one wave per chunk (a 62 k-point view alone: index build 0.053 -> 0.043 ms;
                                         // 8 x 200 k points per launch are 2 % SLOWER with the small chunks, 4 x the table traffic)
constexpr int kInvalidBin = kBuckets;    // bin of the points without a cell
constexpr int kTagSlots = 256;

// Orders this wave's LDS accesses for the compiler (no forwarding of a store into a later load that
// another lane may have overwritten, no reordering across it).  The hardware needs nothing: a wave's
// DS instructions execute in order.  NOT __syncthreads(): that also waits for every global load and
// store in flight (s_waitcnt vmcnt(0)), which would serialise the rounds of the sort on HBM latency.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// rank = lanes below this one with the same key among the participating lanes, cnt = their number.
// Fast path: every lane writes its number into a small hashed tag table and reads it back; if every
// participating lane reads its own number no two of them share a hash slot, let alone a key.
// Otherwise one ballot per key bit narrows each lane's set of equals.
__device__ __forceinline__ void match_rank(int key, bool in, int nbits, int *tag, int lane, int &rank, int &cnt) {
    const int hslot = key & (kTagSlots - 1);
    if (in) tag[hslot] = lane;
    wave_lds_fence();
    const bool lost = in && tag[hslot] != lane;
    wave_lds_fence();
    rank = 0;
    cnt = 1;
    if (__any(lost)) {
        unsigned long long m = __ballot(in);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = (key >> bit) & 1;
            const unsigned long long bb = __ballot(in && one);
            m &= one ? bb : ~bb;
        }
        rank = __popcll(m & ((1ull << lane) - 1ull));
        cnt = __popcll(m);
    }
}

__device__ __forceinline__ int point_cell(const GridDesc &g, const char *xyz, size_t stride, int i) {
    const float *p = point_at(xyz, stride, i);
    const float x = p[0], y = p[1], z = p[2];
    if (g.ncells == 0 || !finite3(x, y, z)) return -1;
    const int cx = cell_coord(x, g.mn[0], g.h, g.dims[0]);
    const int cy = cell_coord(y, g.mn[1], g.h, g.dims[1]);
    const int cz = cell_coord(z, g.mn[2], g.h, g.dims[2]);
    return (cz * g.dims[1] + cy) * g.dims[0] + cx;
}

__host__ __device__ inline int sort_chunks(int n, int chunk_pts) { return (n + chunk_pts - 1) / chunk_pts; }

constexpr int kBins = kBuckets + 1;      // buckets + the bin of the points without a cell
constexpr int kRoundsAhead = 8;          // rounds of a chunk whose loads are issued together

// btable[bin * nchunks + chunk] = points of the chunk in the bin (a bin's chunks are contiguous: the
// two kernels that turn counts into offsets give each bin to one wave); cid[i] = cell of point i (-1: none)
__global__ __launch_bounds__(kWave) void bucket_hist_kernel(Batch b, int chunk_pts) {
    const ViewDev &v = b.view[blockIdx.y];
    const int n = v.n, chunk = blockIdx.x, i0 = chunk * chunk_pts, lane = threadIdx.x;
    if (i0 >= n) return;
    __shared__ int hist[kBins];
    for (int k = lane; k < kBins; k += kWave) hist[k] = 0;
    __syncthreads();
    const GridDesc g = v.ds->grid;
    const int bshift = v.ds->bshift;
    for (int r0 = 0; r0 < chunk_pts / kWave; r0 += kRoundsAhead) {
        int c[kRoundsAhead];
#pragma unroll
        for (int k = 0; k < kRoundsAhead; ++k) {
            const int i = i0 + (r0 + k) * kWave + lane;
            c[k] = i < n ? point_cell(g, v.xyz, v.xs, i) : -2;
        }
#pragma unroll
        for (int k = 0; k < kRoundsAhead; ++k) {
            const int i = i0 + (r0 + k) * kWave + lane;
            if (c[k] != -2) {
                v.cid[i] = c[k];
                atomicAdd(&hist[c[k] < 0 ? kInvalidBin : (c[k] >> bshift)], 1);
            }
        }
    }
    __syncthreads();
    const int nchunks = sort_chunks(n, chunk_pts);
    for (int k = lane; k < kBins; k += kWave) v.btable[(size_t)k * nchunks + chunk] = hist[k];
}

// points per bin over all chunks: one wave per bin
__global__ __launch_bounds__(256) void bucket_total_kernel(Batch b, int chunk_pts) {
    const ViewDev &v = b.view[blockIdx.y];
    const int bin = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    if (bin >= kBins) return;
    const int nchunks = sort_chunks(v.n, chunk_pts);
    const int *t = v.btable + (size_t)bin * nchunks;
    int sum = 0;
    for (int c = lane; c < nchunks; c += kWave) sum += t[c];
    for (int off = kWave / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) v.btotal[bin] = sum;
}

// btotal[bin] -> bstart[bin] = first position of the bin; btable[bin][chunk] -> first position of the
// chunk's points inside the bin.  bstart[kBuckets] = number of points with a cell.  Every block
// scans the kBins totals for itself (they are few); then one wave per bin scans the bin's chunks.
// (Totals and starts are separate arrays: the blocks of this launch read ALL totals and finish at
// different times.)
// T threads per workgroup: 256 for one view alone (the shortest chain), 64 for a batch (see kScanBlock)
template <int T>
__global__ __launch_bounds__(T) void bucket_offsets_kernel(Batch b, int chunk_pts) {
    const ViewDev &v = b.view[blockIdx.y];
    __shared__ int tot[kBins];
    __shared__ int wsum[T / kWave];
    for (int k = threadIdx.x; k < kBins; k += blockDim.x) tot[k] = v.btotal[k];
    __syncthreads();
    // exclusive scan of tot[] by the block: T threads x ceil(kBins / T) consecutive bins each
    constexpr int kPer = (kBins + T - 1) / T;
    const int first = threadIdx.x * kPer;
    int mine = 0;
    for (int k = 0; k < kPer; ++k) mine += first + k < kBins ? tot[first + k] : 0;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    int incl = mine;
    for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) wsum[wid] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < wid; ++w) run += wsum[w];
    __syncthreads();
    for (int k = 0; k < kPer; ++k)
        if (first + k < kBins) {
            const int x = tot[first + k];
            tot[first + k] = run;
            run += x;
        }
    __syncthreads();
    const int bin = blockIdx.x * (blockDim.x / kWave) + wid;
    if (bin >= kBins) return;
    const int nchunks = sort_chunks(v.n, chunk_pts);
    int *t = v.btable + (size_t)bin * nchunks;
    int pos = tot[bin];
    for (int c0 = 0; c0 < nchunks; c0 += kWave) {
        const int c = c0 + lane;
        const int x = c < nchunks ? t[c] : 0;
        int inc = x;
        for (int off = 1; off < kWave; off <<= 1) {
            const int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (c < nchunks) t[c] = pos + inc - x;
        pos += __shfl(inc, kWave - 1);
    }
    if (lane == 0) v.bstart[bin] = tot[bin];
}

// rec[2 * position] = (x, y, z, index), rec[2 * position + 1] = (nx, ny, nz, cell): positions inside a bin in index order
__global__ __launch_bounds__(kWave) void bucket_scatter_kernel(Batch b, int chunk_pts) {
    const ViewDev &v = b.view[blockIdx.y];
    const int n = v.n, chunk = blockIdx.x, i0 = chunk * chunk_pts, lane = threadIdx.x;
    if (i0 >= n) return;
    __shared__ int run[kBins];
    __shared__ int tag[kTagSlots];
    const int bshift = v.ds->bshift;
    const int nchunks = sort_chunks(n, chunk_pts);
    for (int k = lane; k < kBins; k += kWave) run[k] = v.btable[(size_t)k * nchunks + chunk];
    __syncthreads();
    for (int r0 = 0; r0 < chunk_pts / kWave; r0 += kRoundsAhead) {
        int c[kRoundsAhead];
        float px[kRoundsAhead], py[kRoundsAhead], pz[kRoundsAhead], nx[kRoundsAhead], ny[kRoundsAhead], nz[kRoundsAhead];
#pragma unroll
        for (int k = 0; k < kRoundsAhead; ++k) {            // unconditional, clamped loads: all in flight together
            const int i = i0 + (r0 + k) * kWave + lane, ic = min(i, n - 1);
            const int cell = v.cid[ic];
            c[k] = i < n ? cell : -2;
            const float *p = point_at(v.xyz, v.xs, ic);
            const float *q = point_at(v.nrmsrc, v.ns, ic);
            px[k] = p[0]; py[k] = p[1]; pz[k] = p[2];
            nx[k] = q[0]; ny[k] = q[1]; nz[k] = q[2];
        }
#pragma unroll
        for (int k = 0; k < kRoundsAhead; ++k) {
            const int i = i0 + (r0 + k) * kWave + lane;
            const bool in = c[k] != -2;
            const int key = c[k] < 0 ? kInvalidBin : (c[k] >> bshift);
            int rank, cnt;
            match_rank(key, in, 11, tag, lane, rank, cnt);
            if (in) {
                const int pos = run[key] + rank;
                v.rec[2 * pos] = make_float4(px[k], py[k], pz[k], __int_as_float(i));       // one 32-byte sector
                v.rec[2 * pos + 1] = make_float4(nx[k], ny[k], nz[k], __int_as_float(c[k]));
                if (c[k] < 0) v.pos_of[i] = -1;
            }
            if (in && rank == cnt - 1) run[key] += cnt;      // after every lane's read above (one wave: in order)
            wave_lds_fence();
        }
    }
}

// level 2, one workgroup of kSortWaves waves per bucket.  The bucket's records are cut into
// kSortWaves consecutive pieces (index order), one per wave; every wave counts its piece per cell
// (cnt[wave][cell] in LDS), the workgroup turns the counts into first positions per (cell, wave)
// and writes cell_start[], then every wave walks its piece once more in index order: position =
// its first position of the cell + points of that cell it has seen so far.  A bucket in a dense
// part of the cloud holds thousands of points: with one wave per bucket the longest bucket set the
// time of the whole kernel.
//   LDS: cnt[kSortWaves][lds_cells] + tag[kSortWaves][kTagSlots]; a bucket of more than lds_cells
//   cells (huge grids) is swept in slices of lds_cells cells.
constexpr int kSortWaves = 4;
constexpr int kCellsPerThread = 4;   // of the workgroup's scan: kSortWaves * 64 * kCellsPerThread >= lds_cells (<= 1024)

__global__ __launch_bounds__(kSortWaves *kWave) void cell_sort_store_kernel(Batch b, int lds_cells, int nbits) {
    extern __shared__ int lds[];
    __shared__ int wsum[kSortWaves];
    const ViewDev &v = b.view[blockIdx.y];
    const DevState *ds = v.ds;
    const GridDesc g = ds->grid;
    const int bk = blockIdx.x, lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    const int nb = ds->nbuckets, bshift = ds->bshift;
    int *cell_start = v.cell_start;
    if (g.ncells == 0) {
        if (bk == 0 && threadIdx.x == 0) cell_start[0] = 0;
        return;
    }
    if (bk >= nb) return;
    int *cnt = lds + wid * lds_cells;                               // this wave's counters
    int *tag = lds + kSortWaves * lds_cells + wid * kTagSlots;
    const int nfinite = v.bstart[kBuckets];
    const int s0 = v.bstart[bk], s1 = v.bstart[bk + 1];
    if (bk == nb - 1 && threadIdx.x == 0) cell_start[g.ncells] = nfinite;
    if (s0 == s1) {            // empty bucket (most of a sparse grid): every cell of it starts where the bucket does
        const int c0 = bk << bshift, c1 = min(c0 + (1 << bshift), g.ncells);
        for (int c = c0 + (int)threadIdx.x; c < c1; c += blockDim.x) cell_start[c] = s0;
        return;
    }
    // this wave's piece [w0, w1): whole rounds of 64 records
    const int piece = ((s1 - s0 + kSortWaves * kWave - 1) / (kSortWaves * kWave)) * kWave;
    const int w0 = min(s0 + wid * piece, s1), w1 = min(w0 + piece, s1);
    const int c_lo = bk << bshift;
    const int c_hi = min(c_lo + (1 << bshift), g.ncells);
    const float4 *rec = v.rec;
    float4 *pts_out = v.pts, *nrm_out = v.nrm;
    int sbase = s0;
    for (int sub = c_lo; sub < c_hi; sub += lds_cells) {
        const int m = min(lds_cells, c_hi - sub);
        for (int k = lane; k < m; k += kWave) cnt[k] = 0;
        wave_lds_fence();
        for (int tb = w0; tb < w1; tb += kRoundsAhead * kWave) {
            int cc[kRoundsAhead];
#pragma unroll
            for (int k = 0; k < kRoundsAhead; ++k) {        // unconditional loads, clamped: all in flight together
                const int t = tb + k * kWave + lane;
                const int y = __float_as_int(rec[2 * min(t, w1 - 1) + 1].w);
                cc[k] = t < w1 ? y - sub : -1;
            }
#pragma unroll
            for (int k = 0; k < kRoundsAhead; ++k)
                if (cc[k] >= 0 && cc[k] < m) atomicAdd(&cnt[cc[k]], 1);
        }
        __syncthreads();
        // per cell: counts of the waves -> exclusive prefix over the waves; total per cell
        int tot[kCellsPerThread];
        int mine = 0;
#pragma unroll
        for (int j = 0; j < kCellsPerThread; ++j) {
            const int k = threadIdx.x * kCellsPerThread + j;
            tot[j] = 0;
            if (k < m) {
                int run = 0;
#pragma unroll
                for (int w = 0; w < kSortWaves; ++w) {
                    const int x = lds[w * lds_cells + k];
                    lds[w * lds_cells + k] = run;
                    run += x;
                }
                tot[j] = run;
            }
            mine += tot[j];
        }
        // exclusive scan of the cell totals over the workgroup (kCellsPerThread consecutive cells per thread)
        int incl = mine;
        for (int off = 1; off < kWave; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) wsum[wid] = incl;
        __syncthreads();
        int first = incl - mine, slice_total = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            first += w < wid ? wsum[w] : 0;
            slice_total += wsum[w];
        }
#pragma unroll
        for (int j = 0; j < kCellsPerThread; ++j) {
            const int k = threadIdx.x * kCellsPerThread + j;
            if (k < m) {
                cell_start[sub + k] = sbase + first;                // first position of cell k
#pragma unroll
                for (int w = 0; w < kSortWaves; ++w) lds[w * lds_cells + k] += first;
            }
            first += tot[j];
        }
        __syncthreads();
        // kRoundsAhead rounds at a time: all their records are requested together, then the rounds are
        // ranked and stored one after the other (a wave's loads and stores retire in order: a round
        // that waits for its own loads also waits for the stores of the round before)
        for (int tb = w0; tb < w1; tb += kRoundsAhead * kWave) {
            // (scalar arrays: an array of float4 indexed in an unrolled loop is left in scratch memory)
            float ax[kRoundsAhead], ay[kRoundsAhead], az[kRoundsAhead], aw[kRoundsAhead];
            float bx[kRoundsAhead], by[kRoundsAhead], bz[kRoundsAhead];
            int bc[kRoundsAhead];
#pragma unroll
            for (int k = 0; k < kRoundsAhead; ++k) {
                const int t = min(tb + k * kWave + lane, w1 - 1);
                const float4 a4 = rec[2 * t], b4 = rec[2 * t + 1];
                ax[k] = a4.x; ay[k] = a4.y; az[k] = a4.z; aw[k] = a4.w;
                bx[k] = b4.x; by[k] = b4.y; bz[k] = b4.z; bc[k] = __float_as_int(b4.w);
            }
#pragma unroll
            for (int k = 0; k < kRoundsAhead; ++k) {
                const int cc = bc[k] - sub;
                const bool in = tb + k * kWave + lane < w1 && cc >= 0 && cc < m;
                int rank, same;
                match_rank(cc, in, nbits, tag, lane, rank, same);
                if (in) {
                    const int pos = sbase + cnt[cc] + rank;
                    pts_out[pos] = make_float4(ax[k], ay[k], az[k], aw[k]);
                    // w: the normal is finite; x of one that is not: NaN (all a 12-byte read of the record needs)
                    const bool fin = finite3(bx[k], by[k], bz[k]);
                    nrm_out[pos] = make_float4(fin ? bx[k] : NAN, by[k], bz[k], fin ? 1.0f : 0.0f);
                }
                if (in && rank == same - 1) cnt[cc] += same;   // after every lane's read above (one wave: in order)
                wave_lds_fence();
            }
        }
        sbase += slice_total;
        __syncthreads();
    }
}

// pos_of[original index] = storage position (the points without a cell got -1 from the scatter)
__global__ __launch_bounds__(256) void pos_of_kernel(Batch b) {
    const ViewDev &v = b.view[blockIdx.y];
    if (!v.want_pos_of) return;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= v.n || s >= v.cell_start[v.ds->grid.ncells]) return;
    v.pos_of[__float_as_int(v.pts[s].w)] = s;
}

// ---------------------------------------------------------------------------------------------
// Soft assignment, /root/reference/src/KeypointLearning.cpp:41-65 and :68-92: soft_pair.h (soft_pair, clamp_cosine
// and the per-launch constants dim, dim/2, RN(1/dim)) -- one definition shared with tools/check_soft_pair.hip, which
// runs it on the device over the table generated by the reference's own functions (tests/golden/pair_kat.json).
// div_rn(a, b, rb) = the correctly rounded quotient a / b in 3 instructions, sqrt_rn(x) = the correctly rounded
// square root in 9: exact_math.h.
// ---------------------------------------------------------------------------------------------
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

// FLANN L2_Simple<float>: d = dx*dx; d += dy*dy; d += dz*dz.  x and y go through the packed
// (2 x f32) add / multiply of gfx950 as ONE pair per candidate -- the same IEEE operations in the
// same order; left to itself the compiler pairs components of two different candidates and pays
// for it in register moves.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
template <class Q>
__device__ __forceinline__ float dist2(float px, float py, float pz, const Q &q) {
    const f32x2 pxy = {px, py}, qxy = {q.x, q.y};
    const f32x2 dxy = pxy - qxy;
    const f32x2 sq = dxy * dxy;
    const float dz = pz - q.z;
    float d = sq.x;
    d += sq.y;
    d += dz * dz;
    asm("" : "+v"(d));   // keeps the vectorizer from pairing this chain with another candidate's
    return d;
}

constexpr int kLanes = 64;   // lanes of a wave = threads of a workgroup of the feature kernels

// what one accepted neighbor adds to the histogram: 4 cells and 4 weights (hpp:342-355).  The
// cells are byte offsets of the lane's entries from H (cell c of lane l lives at (c * 64 + l) * 4)
struct Contribution {
    int c0, c1, c2, c3;      // (a,b) (a,b') (a',b) (a',b'): LDS addresses
    float w00, w01, w10, w11;
};

// kCols = points whose histograms share the LDS block of a wave (H[c * kCols + col]): 64 / kGroup
template <int kCols, class NQ>
__device__ __forceinline__ Contribution neighbor_contribution(const FeatDesc &f, float d2,
                                                              const float4 &np, const NQ &nq, int col_address) {
    const float dot = np.x * nq.x + (np.y * nq.y + np.z * nq.z);                   // hpp:342
    float cosine = 1 - dot;
    int a, ap, bi, bp;
    float aw, bw;
    soft_pair(f.A, f.A1f, sqrt_rn(d2), f.ann_dim, f.ann_half, f.ann_rdim, a, ap, aw);   // hpp:345
    cosine = clamp_cosine(cosine);                                                 // cpp:70-73
    soft_pair(f.B, f.B1f, cosine, f.bin_dim, f.bin_half, f.bin_rdim, bi, bp, bw);          // hpp:348
    Contribution c;
    c.w00 = (1 - bw) * (1 - aw);
    c.w01 = bw * (1 - aw);
    c.w10 = (1 - bw) * aw;
    c.w11 = bw * aw;
    const int row_bytes = f.B * (kCols * 4), lane_bytes = col_address;      // LDS address of H[0 * kCols + point]
    const int ra = __mul24(a, row_bytes) + lane_bytes, rap = __mul24(ap, row_bytes) + lane_bytes;
    const int cb = bi * (kCols * 4), cbp = bp * (kCols * 4);
    c.c0 = ra + cb;
    c.c1 = ra + cbp;
    c.c2 = rap + cb;
    c.c3 = rap + cbp;
    return c;
}

// LDS by address: the cells of a Contribution are LDS ADDRESSES (the base of H is folded into the point's
// column address once per kernel: added per access it was four `v_add_u32 v, 0, v` per round, the base
// being a link-time 0 the compiler cannot fold)
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ int lds_address(const float *p) { return (int)(size_t)(const lds_float *)p; }
__device__ __forceinline__ lds_float &hist_at(int lds_byte_address) {
    return *(lds_float *)(size_t)(unsigned)lds_byte_address;
}

// The 4 "+=" of one neighbor (hpp:350-355) may hit the same cell -- the pair index is clamped onto
// the index at the range ends, the common case for bin 0.  The 4 cells are read once
// (request_cells), the adds are forwarded through registers in the reference's order and written
// back in order (apply_contribution), so the float result is exactly the one the sequential chain
// gives and only one LDS round trip sits on the critical path.
// Which cells coincide follows from two facts: c1 == c0 and c3 == c2 iff the bin pair was clamped onto the
// bin, c2 == c0 and c3 == c1 iff the annulus pair was clamped onto the annulus.
// (Four ds_add_f32 give the same bits -- the LDS adder rounds like v_add_f32 and one wave's DS
// instructions execute in order -- but LDS float atomics run at a fraction of the plain read /
// write rate: the kernel took 2.3x as long with them, profiles/r02_notes.md.)
struct Cells {
    float v0, v1, v2, v3;
};

__device__ __forceinline__ Cells request_cells(const Contribution &c) {
    Cells v;
    v.v0 = hist_at(c.c0);
    v.v1 = hist_at(c.c1);
    v.v2 = hist_at(c.c2);
    v.v3 = hist_at(c.c3);
    return v;
}

__device__ __forceinline__ void apply_contribution(const Contribution &c, const Cells &v) {
    // (the clamp flags, re-derived from the cells here: carried as booleans across the branch around the
    // contribution they cost eight instructions per round to unpack)
    const bool same_b = c.c1 == c.c0, same_a = c.c2 == c.c0;
    const bool both = same_a & same_b;
    const float x0 = v.v0 + c.w00;                                                 // hpp:350
    const float x1 = (same_b ? x0 : v.v1) + c.w01;                               // hpp:351
    const float x2 = (both ? x1 : same_a ? x0 : v.v2) + c.w10;                   // hpp:354
    const float x3 = (same_b ? x2 : same_a ? x1 : v.v3) + c.w11;               // hpp:355
    hist_at(c.c0) = x0;
    hist_at(c.c1) = x1;
    hist_at(c.c2) = x2;
    hist_at(c.c3) = x3;
}

// 12-byte / 4-byte loads addressed by a 32-bit byte offset from a wave-uniform base (one shift
// instead of 64-bit address arithmetic per load; views are limited to 2^28 points).
// The first 12 bytes of a 16-byte record: one dwordx3 load (a quarter less for the texture path to return)
__device__ __forceinline__ f32x3 ld12(const float4 *__restrict__ base, int idx) {
    return *reinterpret_cast<const f32x3 *>(reinterpret_cast<const char *>(base) + ((unsigned)idx << 4));
}
__device__ __forceinline__ int ld4(const int *__restrict__ base, int idx) {
    return *reinterpret_cast<const int *>(reinterpret_cast<const char *>(base) + ((unsigned)idx << 2));
}

// a wave-uniform value as a scalar-register VALUE (v_readfirstlane), not a reloadable kernel argument
__device__ __forceinline__ int pin_i(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float pin_f(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

constexpr int kStepW = 4;      // candidates per search step: one address, kStepW 16-byte loads

struct Cand {
    f32x3 q[kStepW];
};

// the kStepW candidates starting at storage position t: ONE address, constant offsets.  Positions
// past the end of the row hold other cells' points (or, past the last point, the kStepW padding
// elements of the array, pts_bytes()); their accept bits are masked by the row end
__device__ __forceinline__ Cand load_cand(const float4 *__restrict__ pts, int t) {
    Cand c;
#pragma unroll
    for (int j = 0; j < kStepW; ++j) c.q[j] = ld12(pts + j, t);   // uniform base + j, one 32-bit offset
    return c;
}

// One search step: the accept bits of kStepW candidates are shifted into w from the right, so the
// first candidate of a word ends up in its highest bit.  Strict d2 < r2 (KdTreeFLANN::radiusSearch)
// as the sign of RN(d2 - r2): the difference of two floats is zero only if they are equal
// (denormals are kept), so the sign bit is set exactly when d2 < r2.
__device__ __forceinline__ unsigned search_step(unsigned w, const float4 &p, const Cand &c, float r2) {
#pragma unroll
    for (int j = 0; j < kStepW; ++j) {
        const float s = dist2(p.x, p.y, p.z, c.q[j]) - r2;
        w = __builtin_amdgcn_alignbit(w, __float_as_uint(s), 31);   // (w << 1) | sign(s)
    }
    return w;
}

// ---------------------------------------------------------------------------------------------
// computePointFeatures, hpp:321-376: G = kGroup lanes per query point (lanes G i .. G i + G - 1 of the
// wave: a "group"), 64 / G consecutive storage positions per wave (spatially coherent: same or adjacent
// cells); one wave per workgroup, waves never synchronise with each other.  Returns K_f.
//
// Two alternating phases per wave:
//   search  the group walks the rows of cells of its point's search box in canonical order ((cz, cy)
//           ascending, storage positions ascending) TOGETHER: 4 G candidates per step (4 per lane, adjacent
//           pieces of the row), requested one step earlier; the 4-bit results (sign of d2 - r2 through
//           v_alignbit) are OR-ed across the group with DPP, 8 / G steps make an accept word of 32 consecutive
//           candidates; non-empty words go to the POINT's list in LDS.  The first accepted neighbor of the
//           query is dropped here (hpp:336 starts at neigh_indx = 1).  All points of the wave are at the same
//           row slot of their boxes; a row costs the wave as many steps as its longest instance.
//   drain   G neighbors of the point per round, in order, from the list of words: lane g takes the g-th set
//           bit of what is left of the current word (a round never spans two words), requests point and
//           normal one round ahead of their use, and computes its contribution -- exact sqrt, two soft
//           assignments: the expensive, order-free part.  Then the G contributions are added to the point's
//           histogram one lane after the other (4 cells read, added with register forwarding where cells
//           coincide, written back): the order of hpp:350-355 over the neighbors is the order of the lanes
//           (LDS operations of a wave execute in order).
// The search runs until a point's list is full (ecap words) or the rows are used up; the lists are then
// drained and the search resumes.  Neighbors never go through global memory; nothing but the histograms
// and the accept words lives in LDS:
//   H[c * (64 / G) + point]     float  the point's A x B histogram
//   ent[e * (64 / G) + point]   uint2  its e-th non-empty accept word: x = storage position of the word's first
//                                      candidate, y = accept bits (first candidate = highest bit)
// One lane per point (rounds 1 and 2 until r02e) needs fewer instructions per neighbor (no bit selection,
// one add step instead of G: x 1.4 at G = 2, x 1.8 at G = 4) but twice the LDS per wave, runs every loop for
// the maximum over 64 instead of 32 points, and moves a point half as fast: G = 2 measured faster on every
// workload -- 8 views of 200 k points 0.667 -> 0.652 ms, one such view 0.137 -> 0.118, one 62 k-point
// view 0.091 -> 0.076, 500 k points at r = 10 mr 0.767 -> 0.494, config 5 (F = 80) 1.72 -> 1.21 ms; G = 4:
// 0.164 / 0.088 / - / 1.19 (profiles/r02_notes.md).
constexpr int kGroup = 2;

template <int G>
__host__ __device__ inline size_t feature_lds_bytes(int F, int ecap) {
    return (sizeof(float) * (size_t)F + sizeof(uint2) * (size_t)ecap) * (size_t)(kLanes / G);
}

// OR over the G lanes of a group (DPP quad_perm [1,0,3,2], then [2,3,0,1], then row_half_mirror: lane i <-> 7 - i of
// every 8 lanes, which pairs the two quads of the group)
template <int G>
__device__ __forceinline__ unsigned group_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);
    if (G >= 4) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);
    if (G == 8) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);
    return v;
}

// x without its highest set bit (0 stays 0)
__device__ __forceinline__ unsigned drop_first_bit(unsigned x) { return x & (0x7fffffffu >> (__clz((int)x) & 31)); }
// The accept words are BUILT with the first candidate in the highest bit (the sign bits are shifted in from the
// right) and STORED bit-reversed, first candidate = bit 0: taking the next neighbor off a word is then x & (x - 1)
// and its position a count of trailing zeros -- two full-rate instructions instead of four half-rate ones per taken
// bit (tools/valu_ceiling.hip: v_add_u32 / v_and_b32 issue in 2.4 cycles, v_ffbh / v_min / v_lshrrev in 4.2).
__device__ __forceinline__ unsigned drop_lowest_bit(unsigned x) { return x & (x - 1u); }
// position of the lowest set bit of m (any value when m == 0: the callers discard it)
__device__ __forceinline__ int lowest_bit_index(unsigned m) { return __builtin_ctz(m | 0x80000000u); }

template <int G>
__device__ __forceinline__ int point_features(const float4 *__restrict__ pts, const float4 *__restrict__ nrm,
                                                    const int *__restrict__ cell_start, const GridDesc &g,
                                                    const FeatDesc &fin, float4 p, float4 np, float *H, uint2 *ent,
                                                    int ecap, bool active, float *raw_mass = nullptr) {
    static_assert(G == 2 || G == 4, "lanes per point");
    constexpr int kPts = kLanes / G, kStepBits = kStepW * G, kSteps = 32 / kStepBits;    // 16 / 8 candidates per step
    const int tid = threadIdx.x, pi = tid / G, gq = tid % G;
    // the per-launch constants of the loops below, pinned in scalar registers: left as kernel arguments the
    // compiler re-reads them from memory inside the accumulate loop (scalar loads whose s_waitcnt lgkmcnt(0)
    // also waits for the LDS reads in flight)
    FeatDesc f;
    f.A = pin_i(fin.A);
    f.B = pin_i(fin.B);
    f.F = pin_i(fin.F);
    f.A1f = pin_f(fin.A1f);
    f.B1f = pin_f(fin.B1f);
    f.support = fin.support;
    f.ann_dim = pin_f(fin.ann_dim);
    f.ann_half = pin_f(fin.ann_half);
    f.ann_rdim = pin_f(fin.ann_rdim);
    f.bin_dim = pin_f(fin.bin_dim);
    f.bin_half = pin_f(fin.bin_half);
    f.bin_rdim = pin_f(fin.bin_rdim);
    f.r2 = pin_f(fin.r2);
    f.rr = fin.rr;
    for (int c = gq; c < f.F; c += G) H[c * kPts + pi] = 0.0f;                     // hpp:325
    CellBox b = make_box(g, p.x, p.y, p.z, f.rr);
    b.hi[1] = min(b.hi[1], b.lo[1] + 3);
    b.hi[2] = min(b.hi[2], b.lo[2] + 3);
    const int ny = active ? b.hi[1] - b.lo[1] + 1 : 0, nz = active ? b.hi[2] - b.lo[2] + 1 : 0;
    const int wny = __any(ny > 3) ? 4 : __any(ny > 2) ? 3 : __any(ny > 1) ? 2 : __any(ny > 0) ? 1 : 0;
    const int wnz = __any(nz > 3) ? 4 : __any(nz > 2) ? 3 : __any(nz > 1) ? 2 : __any(nz > 0) ? 1 : 0;
    const int t_max = max(cell_start[g.ncells] - 1, 0);
    auto row_range = [&](int ky, int kz, int &r0, int &r1) {
        const bool valid = (ky < ny) & (kz < nz);
        const int row = valid ? ((b.lo[2] + kz) * g.dims[1] + b.lo[1] + ky) * g.dims[0] : 0;
        const int x = ld4(cell_start, row + (valid ? b.lo[0] : 0));
        const int y = ld4(cell_start, row + (valid ? b.hi[0] + 1 : 0));
        r0 = valid ? x : 0;
        r1 = valid ? y : 0;
    };
    int ky = 0, kz = 0;
    bool slots_left = wny > 0 && wnz > 0;
    int n0 = 0, n1 = 0;
    if (slots_left) row_range(0, 0, n0, n1);
    int t = 0, t1 = 0;                            // current row of the group: next candidate of lane 0, end
    const int mine = kStepW * gq;                 // this lane's 4 candidates of a step start here
    Cand pre = load_cand(pts, 0);
    bool first_pending = true;
    int kf = 0;
    const int ent_last = (ecap - 1) * kPts + pi;
    const int nib_shift = 4 * (G - 1 - gq);
    const int col_address = lds_address(H + pi);
    for (;;) {
        // ================= search: accept words of the point (identical in the lanes of the group) =================
        int ecnt = 0;
        bool full = false;
        while (!full) {
            if (!__any(t < t1)) {
                if (!slots_left) break;
                t = n0;
                t1 = n1;
                pre = load_cand(pts, min(t + mine, t_max));
                if (++ky == wny) {
                    ky = 0;
                    ++kz;
                }
                slots_left = kz < wnz;
                if (slots_left) row_range(ky, kz, n0, n1);
                continue;
            }
            // one word = kSteps steps of 4 G candidates, two per round so that the two candidate sets swap roles
            const int wbase = t;
            unsigned w = 0u;
#pragma unroll
            for (int r = 0; r < kSteps / 2; ++r) {
                Cand nxt = load_cand(pts, min(t + kStepBits + mine, t_max));
                w = (w << kStepBits) | group_or<G>(search_step(0u, p, pre, f.r2) << nib_shift);
                pre = load_cand(pts, min(t + 2 * kStepBits + mine, t_max));
                w = (w << kStepBits) | group_or<G>(search_step(0u, p, nxt, f.r2) << nib_shift);
                t += 2 * kStepBits;
                if (r + 1 < kSteps / 2 && !__any(t < t1)) {              // the row is over for every point: a short word
                    w <<= 32 - 2 * kStepBits * (r + 1);
                    break;
                }
            }
            const int nv = min(max(t1 - wbase, 0), 32);                  // candidates of this word inside the row
            w = nv > 0 ? (w & (0xffffffffu << ((32 - nv) & 31))) : 0u;
            kf += __popc(w);
            if (first_pending & (w != 0u)) {                             // hpp:336
                w = drop_first_bit(w);
                first_pending = false;
            }
            if (w != 0u) {
                if (gq == 0) ent[ecnt * kPts + pi] = make_uint2((unsigned)wbase, __brev(w));     // first candidate = bit 0
                ++ecnt;
            }
            full = __any(ecnt == ecap);
        }
        // ================= drain: G neighbors of the point per round =================
        if (__any(ecnt > 0)) {
            wave_lds_fence();                      // the words were stored by lane 0 of the group
            struct Taken {
                bool valid;
                f32x3 q, n;      // n.x is NaN for a normal that is not finite
            };
            int e = 0;                             // words of the point's list consumed
            unsigned w = 0u;
            int wbase = 0;
            uint2 nw = ent[pi];                    // next word, requested one round ahead
            auto take = [&](Taken &slot) {
                const bool refill = (w == 0u) & (e < ecnt);
                w = refill ? nw.y : w;
                wbase = refill ? (int)nw.x : wbase;
                e += refill ? 1 : 0;
                nw = ent[min(e * kPts + pi, ent_last)];
                // the next G set bits of the word, one per lane (a round never spans two words)
                unsigned m;
                if (G == 2) {
                    const unsigned c1 = drop_lowest_bit(w);
                    m = gq == 0 ? w : c1;
                    w = drop_lowest_bit(c1);
                } else {
                    const unsigned c1 = drop_lowest_bit(w), c2 = drop_lowest_bit(c1), c3 = drop_lowest_bit(c2);
                    m = gq == 0 ? w : gq == 1 ? c1 : gq == 2 ? c2 : c3;
                    w = drop_lowest_bit(c3);
                }
                slot.valid = m != 0u;
                const int tt = slot.valid ? wbase + lowest_bit_index(m) : 0;
                slot.q = ld12(pts, tt);
                slot.n = ld12(nrm, tt);
            };
            Taken pa, pb;
            pa.valid = pb.valid = false;
            pa.q = pa.n = pb.q = pb.n = f32x3{0.f, 0.f, 0.f};
#define KPL_GROUP_ROUND(now, nxt)                                                                  \
    {                                                                                              \
        take(nxt);                                                                                 \
        const bool has_ = now.valid & (now.n.x == now.n.x);                         /* hpp:338 */  \
        Contribution c_;                                                                           \
        if (has_) c_ = neighbor_contribution<kPts>(f, dist2(p.x, p.y, p.z, now.q), np, now.n, col_address); \
        _Pragma("unroll") for (int sub_ = 0; sub_ < G; ++sub_) {                                   \
            if (has_ & (gq == sub_)) apply_contribution(c_, request_cells(c_));                    \
            wave_lds_fence();                                                                      \
        }                                                                                          \
        now.valid = false;                                                                         \
    }
            do {
                KPL_GROUP_ROUND(pa, pb)
                KPL_GROUP_ROUND(pb, pa)
            } while (__any((w != 0u) | (e < ecnt) | pa.valid | pb.valid));
#undef KPL_GROUP_ROUND
        }
        if (!slots_left && !__any(t < t1)) break;
    }
    wave_lds_fence();
    // the sum of the lane's cells BEFORE the rows are normalised: every neighbor with a finite normal added weights that sum
    // to one, so the histogram's mass is the size of the neighborhood (minus the dropped first neighbor) -- the caller's
    // sample of K_f for the handle's next launch (note_kf), taken by the sampled waves only and at no cost to the loops above
    // (kept live through them, the count of accepted neighbors took the kernel from 96 to 98 registers = from 5 waves per
    // SIMD to 4: -2.5 % on the bench)
    if (raw_mass) {
        float m = 0.0f;
        for (int c = gq; c < f.F; c += G) m += H[c * kPts + pi];
        *raw_mass = m;
    }
    for (int a = gq; a < f.A; a += G) {                                            // hpp:360-370, one row per lane
        float *h = H + (a * f.B) * kPts + pi;
        float s = 0.0f;
        for (int k = 0; k < f.B; ++k) {
            float v = h[k * kPts];
            s += v * v;
        }
        const float nr = sqrtf(s);
        if (nr > 0)
            for (int k = 0; k < f.B; ++k) h[k * kPts] = h[k * kPts] / nr;
    }
    wave_lds_fence();
    return kf;
}

// ---------------------------------------------------------------------------------------------
// computePointFeatures for LARGE neighborhoods in TWO PASSES (FeatDesc::walk == kWalkTwoPass; the reference's own default
// operating point: radiusFeatures 20 on the cheff views = 30 mesh resolutions, ~730 points per cell, K_f ~ 2 300, ~10 000
// candidates in the 27 cells around a point -- /root/reference/src/main_test_detector.cpp:65).  Same neighbor set, same
// canonical order, same arithmetic as point_features: the results are the same bits.
//
// point_features alternates search and drain whenever ONE point of the wave has filled its 24 accept words in LDS.  With
// thousands of neighbors per point that is a drain every few hundred candidates, and every drain lasts as long as the
// point with the most accepted neighbors SINCE THE LAST ONE: on cheff001 a wave runs 1 700 drain rounds of two neighbors
// per point where its longest neighborhood needs 1 260, and the waves of the densest cells -- 2.7 x the candidates, 1.5 x
// the neighbors of the average wave -- run 3 600, alone on their SIMDs for the second half of the launch (measured with
// in-kernel stamps: profiles/r05_notes.md).  Here the accept words of a point's WHOLE walk go to a list in global
// memory first (search_point_words_staged: nothing but the search, the candidates staged in LDS once per wave), and a second kernel drains every list
// in one go (drain_point_words): its rounds are those of the wave's longest neighborhood, once.
//   word list of a wave   ViewDev::sort_keys (an array of 8-byte records that the sorted-search mode uses for its keys; a
//                         view is in one mode or the other), one contiguous block per wave of the search kernel:
//                         entry e of the wave's point pi at block + e * (64 / G) + pi -- x = storage position of the
//                         word's first candidate, y = accept bits (first candidate = bit 0), the first accepted neighbor
//                         of the point already dropped (hpp:336).  seg_start[s] = block + pi, seg_len[s] = entries.
//   blocks                bump allocation, one returning atomic per wave on one of 32 cursors (DevState::word_cursor: returning
//                         atomics on one address complete one after the other, ~11 ns each -- 8 000 waves on one cursor
//                         would queue for 90 us), each over a 32nd of the array.  The size of a block is known before the walk: the wave steps
//                         through its rows in lock step, one word per step.  A block that does not fit sets
//                         kStatusKeyCapacity: the call fails with KPL_ERR_RETRY and finds the array grown (kpl_sync_status).
// ---------------------------------------------------------------------------------------------
constexpr int kWalkLanes = 0, kWalkTwoPass = 1;      // FeatDesc::walk
constexpr int kWordShards = 32;

// Pass 1 for the cells that hold MANY points (what the two-pass walk exists for): with hundreds of points per cell the
// 64 / G points of a wave -- consecutive storage positions -- nearly always share ONE cell, hence one search box, and walk
// the same rows.  Were every lane to fetch its candidates itself (as point_features does): 8 twelve-byte loads per word
// and lane, every lane at an address of its own -- the texture addresser of the CU takes them a lane at a time, and a kernel
// that does nothing but search is bound by it (cheff001: 0.53 ms, measured in round 5).  Here the wave fetches the
// candidates ONCE, for all its points:
//   group    the points of the wave that lie in the same grid cell (usually all of them; a wave that straddles cells
//            takes its groups one after the other).  Their boxes differ by a cell at most: the wave walks the UNION of
//            them -- rows (cz, cy) ascending, every row the run of cells lo_x .. hi_x --, a superset of every point's
//            own box in the same order; what lies outside a point's own box is farther than rr > r from it in one
//            coordinate and fails the distance test like any other candidate.
//   stage    the row in windows of W consecutive storage positions, copied to LDS by the whole wave
//            (global_load_lds_dwordx4: 1 KiB per instruction, no registers, all of a window in flight together)
//   search   as in point_features -- 4 G candidates per step, accept words of 32 candidates, the first accepted
//            neighbor of a point dropped (hpp:336) -- but every group reads the SAME candidates, from LDS
// The words of a point are cut at other positions than in point_features (rows start at the union's first cell);
// the neighbors they stand for, and their order, are the same.
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_cvoid;

// minimum / maximum over the lanes of the wave that take part (every lane returns the result)
__device__ __forceinline__ int wave_min_of(int v, bool in) {
    v = in ? v : 0x7fffffff;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wave_max_of(int v, bool in) {
    v = in ? v : (int)0x80000000;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}

template <int G, class Block>
__device__ __forceinline__ int search_point_words_staged(const float4 *__restrict__ pts, const int *__restrict__ cell_start,
                                                         const GridDesc &g, const FeatDesc &fin, float4 p, bool active,
                                                         float4 *sp, int W, Block block, int &entries, bool drop_first = true) {
    static_assert(G == 2 || G == 4 || G == 8, "lanes per point");
    constexpr int kPts = kLanes / G, kStepBits = kStepW * G, kSteps = 32 / kStepBits;
    const int tid = threadIdx.x, pi = tid / G, gq = tid % G;
    const float r2 = pin_f(fin.r2);
    const CellBox b = make_box(g, p.x, p.y, p.z, fin.rr);
    const int cx = cell_coord(p.x, g.mn[0], g.h, g.dims[0]), cy = cell_coord(p.y, g.mn[1], g.h, g.dims[1]),
              cz = cell_coord(p.z, g.mn[2], g.h, g.dims[2]);
    const int own = (cz * g.dims[1] + cy) * g.dims[0] + cx;
    // the union of the boxes of a group.  Its points share their cell c, and a box reaches two cells beyond the point's own at
    // most (rr < 1.001 cell edges): the bounds come out of a few wave votes -- a reduction over the lanes is a chain of six
    // cross-lane moves per bound, and a wave of this kernel lives for a few hundred word steps only
    struct Union {
        int lx, hx, ly, hy, lz, hz;
    };
    auto vote_min = [&](int v, int c, bool in) -> int {
        if (__any(in && v < c - 2)) return pin_i(wave_min_of(v, in));
        return __any(in && v <= c - 2) ? c - 2 : __any(in && v <= c - 1) ? c - 1 : c;
    };
    auto vote_max = [&](int v, int c, bool in) -> int {
        if (__any(in && v > c + 2)) return pin_i(wave_max_of(v, in));
        return __any(in && v >= c + 2) ? c + 2 : __any(in && v >= c + 1) ? c + 1 : c;
    };
    auto group_union = [&](bool in_group, int leader) {
        const int lcx = __builtin_amdgcn_readlane(cx, leader), lcy = __builtin_amdgcn_readlane(cy, leader),
                  lcz = __builtin_amdgcn_readlane(cz, leader);
        Union u;
        u.lx = vote_min(b.lo[0], lcx, in_group);
        u.hx = vote_max(b.hi[0], lcx, in_group);
        u.ly = vote_min(b.lo[1], lcy, in_group);
        u.hy = vote_max(b.hi[1], lcy, in_group);
        u.lz = vote_min(b.lo[2], lcz, in_group);
        u.hz = vote_max(b.hi[2], lcz, in_group);
        return u;
    };
    // the bounds of the rows rb .. rb + 63 of a union (row k = (lz + k / nyr, ly + k % nyr)), one row per lane: ONE round trip
    // for all of them (fetched one after the other as the walk got to them, the 9 - 25 rows of a box cost a wave more than
    // its word steps: 190 k of its 300 k cycles on cheff001)
    auto row_bounds = [&](const Union &u, int rb, int &br0, int &br1) {
        const int nyr = u.hy - u.ly + 1, nrows = nyr * (u.hz - u.lz + 1), k = rb + tid;
        const bool rv = k < nrows;
        const int rowbase = rv ? ((u.lz + k / nyr) * g.dims[1] + u.ly + k % nyr) * g.dims[0] : 0;
        const int x = ld4(cell_start, rowbase + (rv ? u.lx : 0)), y = ld4(cell_start, rowbase + (rv ? u.hx + 1 : 0));
        br0 = rv ? x : 0;
        br1 = rv ? y : 0;
        return nrows;
    };
    // word steps of the wave = those of its longest group (a point stores the words of its own group only)
    int steps = 0;
    {
        unsigned long long todo = __ballot(active);
        while (todo != 0ull) {
            const int leader = pin_i(__builtin_ctzll(todo));
            const bool in_group = active && own == __builtin_amdgcn_readlane(own, leader);
            todo &= ~__ballot(in_group);
            const Union u = group_union(in_group, leader);
            int words = 0;
            for (int rb = 0;; rb += kLanes) {
                int br0, br1;
                const int nrows = row_bounds(u, rb, br0, br1);
                words += (br1 - br0 + 31) >> 5;
                if (rb + kLanes >= nrows) break;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) words += __shfl_xor(words, d);
            steps = max(steps, pin_i(words));
        }
    }
    entries = 0;
    uint2 *list = block(steps);
    if (list == nullptr) return 0;
    list += pi;
    const int mine = kStepW * gq;                 // this lane's 4 candidates of a step start here
    const int nib_shift = 4 * (G - 1 - gq);
    bool first_pending = drop_first;               // (the sorted order drops element 0 of ITS order, after the sort)
    int kf = 0, ecnt = 0;
    unsigned long long todo = __ballot(active);
    while (todo != 0ull) {
        const int leader = pin_i(__builtin_ctzll(todo));
        const bool in_group = active && own == __builtin_amdgcn_readlane(own, leader);
        todo &= ~__ballot(in_group);
        const Union u = group_union(in_group, leader);
      for (int rb = 0;; rb += kLanes) {              // (one pass: a union has 25 rows at most)
        int br0, br1;
        const int nrows = row_bounds(u, rb, br0, br1);
        const int nb = min(kLanes, nrows - rb);
        // the windows of the group's walk, one after the other: rows in order, a row in pieces of W candidates.
        // Two buffers: the NEXT window is on its way to LDS while the current one is searched (a wave that waited for
        // every window spent more time waiting than searching: 81 windows of a crowded cell at 2-3 us each)
        int jrow = -1, w0 = 0, r1 = 0;                // the row jrow of the batch is cut from w0 on; r1 = its end
        auto next_window = [&](int &o_w0, int &o_nw) -> bool {      // wave-uniform; false: no window left
            for (;;) {
                if (jrow >= 0 && w0 < r1) {
                    o_w0 = w0;
                    o_nw = min(W, r1 - w0);
                    w0 += W;
                    return true;
                }
                if (++jrow >= nb) return false;
                w0 = __builtin_amdgcn_readlane(br0, jrow);
                r1 = __builtin_amdgcn_readlane(br1, jrow);
            }
        };
        auto issue = [&](int a_w0, int a_nw, float4 *buf) {
            for (int j = 0; j < a_nw; j += kLanes) {
                const int src = min(a_w0 + j + tid, a_w0 + a_nw - 1);              // (the last piece: lanes past the end re-read its last record)
                __builtin_amdgcn_global_load_lds((global_cvoid *)(pts + src), (lds_void *)(buf + j), 16, 0, 0);
            }
        };
        int c_w0 = 0, c_nw = 0, n_w0 = 0, n_nw = 0, cur = 0;
        bool have = next_window(c_w0, c_nw);
        if (have) issue(c_w0, c_nw, sp);
        while (have) {
            const bool more = next_window(n_w0, n_nw);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the current window has landed ...
            wave_lds_fence();
            if (more) issue(n_w0, n_nw, sp + (cur ^ 1) * W);          // ... the next one is under way while this one is searched
            const float4 *win = sp + cur * W;
            for (int wb = 0; wb < c_nw; wb += 32) {
                unsigned w = 0u;
                if (G == 8) {
                    // lane gq of the group tests the candidates gq, gq + 8, gq + 16, gq + 24 of the word: every read
                    // instruction of the wave then covers 8 CONSECUTIVE records (128 bytes, every bank once; the 4
                    // consecutive candidates per lane of the other walks put lanes 0 and 4 of a group on the same banks).
                    // All 16 bytes of a record: ds_read_b128 moves 256 bytes per clock, the ds_read_b96 the compiler would
                    // pick 96 -- and this kernel is bound by what the LDS delivers (0.55 -> 0.35 ms on cheff001).
                    Cand c;
                    float4 q[kStepW];
#pragma unroll
                    for (int j = 0; j < kStepW; ++j) q[j] = win[wb + 8 * j + gq];
                    static_assert(kStepW == 4, "the four reads of a word are requested together");
                    asm volatile("" : "+v"(q[0].w), "+v"(q[1].w), "+v"(q[2].w), "+v"(q[3].w));      // (one wait for the four of them)
#pragma unroll
                    for (int j = 0; j < kStepW; ++j) c.q[j] = f32x3{q[j].x, q[j].y, q[j].z};
                    // the lane's 4 accept bits (candidate gq + 8 j in bit 3 - j) -> bits 24, 16, 8, 0, then to the places of
                    // its candidates: candidate c of the word in bit 31 - c
                    const unsigned nib = search_step(0u, p, c, r2);
                    w = group_or<G>(((nib * 0x00204081u) & 0x01010101u) << (7 - gq));
                } else {
#pragma unroll
                    for (int st = 0; st < kSteps; ++st) {
                        Cand c;
                        float4 q[kStepW];
#pragma unroll
                        for (int j = 0; j < kStepW; ++j) q[j] = win[wb + st * kStepBits + mine + j];
                        asm volatile("" : "+v"(q[0].w), "+v"(q[1].w), "+v"(q[2].w), "+v"(q[3].w));
#pragma unroll
                        for (int j = 0; j < kStepW; ++j) c.q[j] = f32x3{q[j].x, q[j].y, q[j].z};
                        const unsigned part = group_or<G>(search_step(0u, p, c, r2) << nib_shift);
                        w = (w << (kStepBits & 31)) | part;
                    }
                }
                const int nv = min(c_nw - wb, 32);                               // candidates of this word inside the window
                w &= 0xffffffffu << ((32 - nv) & 31);
                w = in_group ? w : 0u;
                kf += __popc(w);
                if (first_pending & (w != 0u)) {                                 // hpp:336
                    w = drop_first_bit(w);
                    first_pending = false;
                }
                if (w != 0u) {
                    if (gq == 0) list[ecnt * kPts] = make_uint2((unsigned)(c_w0 + wb), __brev(w));     // first candidate = bit 0
                    ++ecnt;
                }
            }
            wave_lds_fence();                                         // (this window has been read: its buffer is the one after the next)
            have = more;
            c_w0 = n_w0;
            c_nw = n_nw;
            cur ^= 1;
        }
        if (rb + kLanes >= nrows) break;
      }
    }
    entries = ecnt;
    return kf;
}

// Pass 2: the feature loop proper (hpp:334-359) over the point's word list, G neighbors per round as in point_features;
// entry e of the point at list[e * stride].  H as in point_features (zeroed here, normalised here).
template <int G>
__device__ __forceinline__ void drain_point_words(const float4 *__restrict__ pts, const float4 *__restrict__ nrm,
                                                  const FeatDesc &fin, float4 p, float4 np, float *H,
                                                  const uint2 *__restrict__ list, int stride, int ecnt) {
    static_assert(G == 2 || G == 4, "lanes per point");
    constexpr int kPts = kLanes / G;
    const int tid = threadIdx.x, pi = tid / G, gq = tid % G;
    FeatDesc f;                                    // (pinned in scalar registers: see point_features)
    f.A = pin_i(fin.A);
    f.B = pin_i(fin.B);
    f.F = pin_i(fin.F);
    f.A1f = pin_f(fin.A1f);
    f.B1f = pin_f(fin.B1f);
    f.support = fin.support;
    f.ann_dim = pin_f(fin.ann_dim);
    f.ann_half = pin_f(fin.ann_half);
    f.ann_rdim = pin_f(fin.ann_rdim);
    f.bin_dim = pin_f(fin.bin_dim);
    f.bin_half = pin_f(fin.bin_half);
    f.bin_rdim = pin_f(fin.bin_rdim);
    f.r2 = pin_f(fin.r2);
    f.rr = fin.rr;
    for (int c = gq; c < f.F; c += G) H[c * kPts + pi] = 0.0f;                     // hpp:325
    const int col_address = lds_address(H + pi);
    wave_lds_fence();
    if (__any(ecnt > 0)) {
        struct Taken {
            bool valid;
            f32x3 q, n;      // n.x is NaN for a normal that is not finite
        };
        const int last = max(ecnt - 1, 0);
        int e = 0;                             // words of the point's list consumed
        unsigned w = 0u;
        int wbase = 0;
        uint2 nw = ecnt > 0 ? list[0] : make_uint2(0u, 0u);                   // next word, requested ahead of its use
        auto take = [&](Taken &slot) {
            const bool refill = (w == 0u) & (e < ecnt);
            w = refill ? nw.y : w;
            wbase = refill ? (int)nw.x : wbase;
            e += refill ? 1 : 0;
            if (refill) nw = list[min(e, last) * stride];
            unsigned m;
            if (G == 2) {
                const unsigned c1 = drop_lowest_bit(w);
                m = gq == 0 ? w : c1;
                w = drop_lowest_bit(c1);
            } else {
                const unsigned c1 = drop_lowest_bit(w), c2 = drop_lowest_bit(c1), c3 = drop_lowest_bit(c2);
                m = gq == 0 ? w : gq == 1 ? c1 : gq == 2 ? c2 : c3;
                w = drop_lowest_bit(c3);
            }
            slot.valid = m != 0u;
            const int tt = slot.valid ? wbase + lowest_bit_index(m) : 0;
            slot.q = ld12(pts, tt);
            slot.n = ld12(nrm, tt);
        };
        Taken pa, pb;
        pa.valid = pb.valid = false;
        pa.q = pa.n = pb.q = pb.n = f32x3{0.f, 0.f, 0.f};
#define KPL_LIST_ROUND(now, nxt)                                                                   \
    {                                                                                              \
        take(nxt);                                                                                 \
        const bool has_ = now.valid & (now.n.x == now.n.x);                         /* hpp:338 */  \
        Contribution c_;                                                                           \
        if (has_) c_ = neighbor_contribution<kPts>(f, dist2(p.x, p.y, p.z, now.q), np, now.n, col_address); \
        _Pragma("unroll") for (int sub_ = 0; sub_ < G; ++sub_) {                                   \
            if (has_ & (gq == sub_)) apply_contribution(c_, request_cells(c_));                    \
            wave_lds_fence();                                                                      \
        }                                                                                          \
        now.valid = false;                                                                         \
    }
        do {
            KPL_LIST_ROUND(pa, pb)
            KPL_LIST_ROUND(pb, pa)
        } while (__any((w != 0u) | (e < ecnt) | pa.valid | pb.valid));
#undef KPL_LIST_ROUND
    }
    wave_lds_fence();
    for (int a = gq; a < f.A; a += G) {                                            // hpp:360-370, one row per lane
        float *h = H + (a * f.B) * kPts + pi;
        float s = 0.0f;
        for (int k = 0; k < f.B; ++k) {
            float v = h[k * kPts];
            s += v * v;
        }
        const float nr = sqrtf(s);
        if (nr > 0)
            for (int k = 0; k < f.B; ++k) h[k * kPts] = h[k * kPts] / nr;
    }
    wave_lds_fence();
}

// ---------------------------------------------------------------------------------------------
// SORTED-search mode (FeatDesc::sorted): computePointFeatures with the neighbors in ascending (squared
// distance, index) order -- what the feature loop (hpp:334-359) sees when the caller has handed a
// pcl::search::KdTree constructed with sorted = true to the inherited pcl::Keypoint::setSearchMethod
// (/root/reference/include/KeypointLearning.h:56): KdTreeFLANN asks FLANN for sorted results, FLANN sorts its
// RadiusResultSet with DistanceIndex::operator< (distance, then index).  Element 0 of that order -- the query
// itself unless a duplicate of it has a lower index -- is dropped by hpp:336.  It is the one neighbor order
// that is defined without FLANN's tree layout, i.e. the one order in which this engine can be compared bit for
// bit with a PCL run.  The neighbor SET is the canonical mode's; only the order of the float additions differs.
//
// Who scores a point (the handle picks per view from what its last launches measured, api.cpp: prepare_detect):
//   up to ~124 neighbors      feature_sorted_kernel (point_features_sorted_view): kSortGroup = 4 lanes per point, 16 points
//                             per wave -- its own search, positions, 32-bit stand-ins, ordered adds
//   ~100 .. ~420 on average   feature_search_kernel + sorted_words_kernel (point_features_sorted_words): 8 lanes per point,
//                             256 or 512 positions
//   beyond, and every point whose order the stand-ins do not decide (equal distances)
//                             sorted_collect_wave_kernel (up to 512 keys, a wave per point) / sorted_collect_kernel (a
//                             workgroup per point) -> 64-bit keys (d2 bits << 32 | original index: d2 >= +0, so the unsigned
//                             order of the keys IS ascending (d2, index)) in global memory -> sorted_add_kernel
//   sparse queries            features_sorted_kernel (point_features_sorted): 64-bit keys in LDS, in windows of d2
// point_features_sorted, per wave and pass:
//   search   as in point_features: accept words of 32 candidates -> the point's small word list in LDS
//   collect  every accepted neighbor's key -> the point's key list in LDS, lcap keys per point
//   sort     bitonic network over the lists IN REGISTERS (sort_key_lists): lane g of the group holds a quarter of the
//            list, comparators across lanes go through DPP
//   add      the keys in order, 4 per round: normal of the neighbor from the caller's array (by original
//            index), contribution from the d2 in the key, the 4 histogram updates lane after lane (hpp:350-355)
// A neighborhood with more than lcap points takes several passes, each over a window [lo, hi) of keys: when a
// list is about to run full, hi is lowered to a pivot inside the window (its middle in d2) and the list filtered;
// the pass then ends with exactly the keys of [lo, hi), they are sorted and added, and the next pass starts at hi.
// ---------------------------------------------------------------------------------------------
constexpr int kSortGroup = 4;
constexpr int kSortWords = 8;        // accept words per point between two collect rounds

template <int G>
__host__ __device__ inline size_t sorted_lds_bytes(int F, int ecap, int lcap) {          // the sparse queries: 64-bit keys
    return (sizeof(float) * (size_t)F + sizeof(uint2) * (size_t)ecap + sizeof(unsigned long long) * (size_t)lcap) * (size_t)(kLanes / G);
}
template <int G>
__host__ __device__ inline size_t sorted_view_lds_bytes(int F, int ecap, int lcap) {     // whole views: positions
    return (sizeof(float) * (size_t)F + sizeof(uint2) * (size_t)ecap + sizeof(unsigned) * (size_t)lcap) * (size_t)(kLanes / G);
}

// the rows of cells of a point's search box, walked by the G lanes of its group together (the search phase of
// point_features as an object: the sorted mode restarts it for every pass)
template <int G>
struct RowSearch {
    static constexpr int kStepBits = kStepW * G, kSteps = 32 / kStepBits;
    const float4 *__restrict__ pts;
    const int *__restrict__ cell_start;
    const GridDesc *g;
    CellBox b;
    int ny, nz, wny, wnz, t_max, mine, nib_shift;
    float4 p;
    float r2;
    int ky, kz, n0, n1, t, t1;
    bool slots_left;
    Cand pre;

    __device__ __forceinline__ void row_range(int ky_, int kz_, int &r0, int &r1) const {
        const bool valid = (ky_ < ny) & (kz_ < nz);
        const int row = valid ? ((b.lo[2] + kz_) * g->dims[1] + b.lo[1] + ky_) * g->dims[0] : 0;
        const int x = ld4(cell_start, row + (valid ? b.lo[0] : 0));
        const int y = ld4(cell_start, row + (valid ? b.hi[0] + 1 : 0));
        r0 = valid ? x : 0;
        r1 = valid ? y : 0;
    }
    __device__ __forceinline__ void init(const float4 *pts_, const int *cell_start_, const GridDesc &g_, float4 p_, float rr,
                                         float r2_, bool active, int gq) {
        pts = pts_;
        cell_start = cell_start_;
        g = &g_;
        p = p_;
        r2 = r2_;
        b = make_box(g_, p.x, p.y, p.z, rr);
        b.hi[1] = min(b.hi[1], b.lo[1] + 3);
        b.hi[2] = min(b.hi[2], b.lo[2] + 3);
        ny = active ? b.hi[1] - b.lo[1] + 1 : 0;
        nz = active ? b.hi[2] - b.lo[2] + 1 : 0;
        wny = __any(ny > 3) ? 4 : __any(ny > 2) ? 3 : __any(ny > 1) ? 2 : __any(ny > 0) ? 1 : 0;
        wnz = __any(nz > 3) ? 4 : __any(nz > 2) ? 3 : __any(nz > 1) ? 2 : __any(nz > 0) ? 1 : 0;
        t_max = max(cell_start[g_.ncells] - 1, 0);
        mine = kStepW * gq;
        nib_shift = 4 * (G - 1 - gq);
        restart();
    }
    __device__ __forceinline__ void restart() {
        ky = kz = 0;
        slots_left = wny > 0 && wnz > 0;
        n0 = n1 = 0;
        if (slots_left) row_range(0, 0, n0, n1);
        t = t1 = 0;
        pre = load_cand(pts, 0);
    }
    __device__ __forceinline__ bool exhausted() const { return !slots_left && !__any(t < t1); }
    // the next 32 candidates of the point's rows: position of the first one and the accept bits (first candidate
    // = highest bit; 0 where the row had ended for this point); false when every point's rows are used up
    __device__ __forceinline__ bool next_word(int &wbase, unsigned &w) {
        while (!__any(t < t1)) {
            if (!slots_left) return false;
            t = n0;
            t1 = n1;
            pre = load_cand(pts, min(t + mine, t_max));
            if (++ky == wny) {
                ky = 0;
                ++kz;
            }
            slots_left = kz < wnz;
            if (slots_left) row_range(ky, kz, n0, n1);
        }
        wbase = t;
        w = 0u;
#pragma unroll
        for (int r = 0; r < kSteps / 2; ++r) {
            Cand nxt = load_cand(pts, min(t + kStepBits + mine, t_max));
            w = (w << kStepBits) | group_or<G>(search_step(0u, p, pre, r2) << nib_shift);
            pre = load_cand(pts, min(t + 2 * kStepBits + mine, t_max));
            w = (w << kStepBits) | group_or<G>(search_step(0u, p, nxt, r2) << nib_shift);
            t += 2 * kStepBits;
            if (r + 1 < kSteps / 2 && !__any(t < t1)) {
                w <<= 32 - 2 * kStepBits * (r + 1);
                break;
            }
        }
        const int nv = min(max(t1 - wbase, 0), 32);
        w = nv > 0 ? (w & (0xffffffffu << ((32 - nv) & 31))) : 0u;
        return true;
    }
};

// Ascending sort of every point's key list (keys[e * kPts + point], cnt <= G * E keys) by the G lanes of its group, in
// REGISTERS: lane g holds the elements g E .. g E + E - 1 (the list is padded with +infinity to N = G E), and the whole
// bitonic network runs on them -- all comparators ascending (the first step of a merge pairs an element with its mirror
// image in the block), comparators inside a lane are a 64-bit compare and four selects on fixed registers, comparators
// across lanes fetch the partner's element with two DPP moves (quad_perm: the lanes of a group are a quad).  No LDS
// traffic, no index arithmetic: ~2.5 k instructions per wave at E = 32 against ~6-16 k for the same network walked
// over the lists in LDS (profiles/r03_notes.md).
template <int MASK>
__device__ __forceinline__ unsigned long long quad_fetch(unsigned long long x) {
    static_assert(MASK >= 1 && MASK <= 3, "partner inside the quad");
    constexpr int ctrl = MASK == 1 ? 0xB1 : MASK == 2 ? 0x4E : 0x1B;      // quad_perm [1,0,3,2] / [2,3,0,1] / [3,2,1,0]
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)x, ctrl, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(x >> 32), ctrl, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}

// the element of lane g ^ MASK of a group of 8 consecutive lanes (a quad, or two quads of one DPP row): 4 = row_shl / row_shr by 4
// under bank masks (lane_xor32<4> further down), 7 = 4 after 3
__device__ __forceinline__ unsigned lane_xor4_u32(unsigned x) {
    const int t = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xf, 0xa, false);
}
template <int MASK>
__device__ __forceinline__ unsigned long long group_fetch(unsigned long long x) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 3 || MASK == 4 || MASK == 7, "partner inside a group of 8 lanes");
    if (MASK <= 3) return quad_fetch<MASK <= 3 ? MASK : 1>(x);
    if (MASK == 7) x = quad_fetch<3>(x);
    return ((unsigned long long)lane_xor4_u32((unsigned)(x >> 32)) << 32) | lane_xor4_u32((unsigned)x);
}

// (the levels of the network are template instantiations, not iterations of a loop over k: with the partner lanes chosen by
// `if constexpr` the compiler sees straight-line code per level -- as one loop with run-time-looking branches the body exceeded
// the unroll threshold once groups of 8 lanes were added, the loop stayed rolled and the 32 keys of a lane went to scratch)
template <int G, int E>
struct KeySort {
    using U = unsigned long long;
    // (the empty asm statements pin every comparator's results in place, in program order: left alone the scheduler
    // overlaps dozens of comparators and needs 250-300 registers for a network that lives in 64)
    static __device__ __forceinline__ void inside(U &a, U &b) {          // a <- min, b <- max
        const bool sw = b < a;
        const U lo = sw ? b : a, hi = sw ? a : b;
        a = lo;
        b = hi;
        asm volatile("" : "+v"(a), "+v"(b));
    }
    static __device__ __forceinline__ U across(U mine, U other, bool upper) {   // the lower lane keeps the minimum
        const bool lt = other < mine;
        U res = (lt != upper) ? other : mine;
        asm volatile("" : "+v"(res));
        return res;
    }
    // the compare-exchange steps of a merge at distances J, J / 2, ... 1 (elements)
    template <int J>
    static __device__ __forceinline__ void steps(U (&r)[E], int gq) {
        if constexpr (J < E) {
#pragma unroll
            for (int a = 0; a < E; ++a)
                if ((a & J) == 0) inside(r[a], r[a + J]);
        } else {                                    // partner lane g ^ (J / E), the same element
            const bool upper = (gq & (J / E)) != 0;
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = across(r[e], group_fetch<J / E>(r[e]), upper);
        }
        if constexpr (J > 1) steps<J / 2>(r, gq);
    }
    // merges sorted blocks of K / 2 into sorted blocks of K: the first step pairs an element with its mirror image in the
    // block (all comparators ascending), then the half-cleaners
    template <int K>
    static __device__ __forceinline__ void level(U (&r)[E], int gq) {
        if constexpr (K <= E) {                     // inside the lane
#pragma unroll
            for (int blk = 0; blk < E; blk += K)
#pragma unroll
                for (int off = 0; off < K / 2; ++off) inside(r[blk + off], r[blk + K - 1 - off]);
        } else {                                    // partner lane g ^ (K / E - 1), its element E - 1 - e
            const bool upper = (gq & (K / E / 2)) != 0;
#pragma unroll
            for (int e = 0; e < E / 2; ++e) {
                const U o1 = group_fetch<K / E - 1>(r[E - 1 - e]), o2 = group_fetch<K / E - 1>(r[e]);
                r[e] = across(r[e], o1, upper);
                r[E - 1 - e] = across(r[E - 1 - e], o2, upper);
            }
        }
        if constexpr (K >= 4) steps<K / 4>(r, gq);
        if constexpr (2 * K <= G * E) level<2 * K>(r, gq);
    }
};

// The same network on 32-bit keys: two instructions per comparator inside a lane (v_min_u32 / v_max_u32) against five on the
// 64-bit keys, two against six across lanes (one DPP move + v_med3_u32 with the bound 0 in the lower lane = the minimum,
// ~0 in the upper lane = the maximum).
template <int MASK>
__device__ __forceinline__ unsigned group_fetch32(unsigned x) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 3 || MASK == 4 || MASK == 7, "partner inside a group of 8 lanes");
    if (MASK <= 3) {
        constexpr int ctrl = MASK == 1 ? 0xB1 : MASK == 2 ? 0x4E : 0x1B;
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, 0xf, 0xf, true);
    }
    if (MASK == 7) x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x1B, 0xf, 0xf, true);
    return lane_xor4_u32(x);
}

template <int G, int E>
struct KeySort32 {
    using U = unsigned;
    static __device__ __forceinline__ void inside(U &a, U &b) {          // a <- min, b <- max
        const U lo = min(a, b), hi = max(a, b);
        a = lo;
        b = hi;
        asm volatile("" : "+v"(a), "+v"(b));
    }
    static __device__ __forceinline__ U across(U mine, U other, U bound) {
        U res;
        asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(res) : "v"(mine), "v"(other), "v"(bound));
        return res;
    }
    template <int J>
    static __device__ __forceinline__ void steps(U (&r)[E], int gq) {
        if constexpr (J < E) {
#pragma unroll
            for (int a = 0; a < E; ++a)
                if ((a & J) == 0) inside(r[a], r[a + J]);
        } else {
            const U bound = (gq & (J / E)) != 0 ? ~0u : 0u;
#pragma unroll
            for (int e = 0; e < E; ++e) r[e] = across(r[e], group_fetch32<J / E>(r[e]), bound);
        }
        if constexpr (J > 1) steps<J / 2>(r, gq);
    }
    template <int K>
    static __device__ __forceinline__ void level(U (&r)[E], int gq) {
        if constexpr (K <= E) {
#pragma unroll
            for (int blk = 0; blk < E; blk += K)
#pragma unroll
                for (int off = 0; off < K / 2; ++off) inside(r[blk + off], r[blk + K - 1 - off]);
        } else {
            const U bound = (gq & (K / E / 2)) != 0 ? ~0u : 0u;
#pragma unroll
            for (int e = 0; e < E / 2; ++e) {
                const U o1 = group_fetch32<K / E - 1>(r[E - 1 - e]), o2 = group_fetch32<K / E - 1>(r[e]);
                r[e] = across(r[e], o1, bound);
                r[E - 1 - e] = across(r[E - 1 - e], o2, bound);
            }
        }
        if constexpr (K >= 4) steps<K / 4>(r, gq);
        if constexpr (2 * K <= G * E) level<2 * K>(r, gq);
    }
};

// The lists are sorted through 32-bit STAND-INS of their keys where that decides the order: a key (d2, index) in slot i of its
// list is represented by  q(d2) << SB | i,  q(d2) = (unsigned)(d2 * scale)  with scale ~ 2^(31 - SB) / r2 -- float multiply
// and truncation are monotonic, so d2_a < d2_b gives q_a <= q_b, and wherever the q of neighbours in the sorted stand-ins
// DIFFER their order is the order of the keys.  On a surface d2 is spread evenly over [0, r2): two of 70 keys share one of 2^24
// values of q once in 7 000 points.  The network runs on the stand-ins (KeySort32), the 64-bit keys are fetched through the slot
// numbers and written back in order.  If ANY two neighbours of ANY list of the wave share a q (equal or almost equal distances:
// a lattice, duplicates) the wave sorts the 64-bit keys themselves (KeySort) -- the exact order either way.
// Padding: slot i >= cnt stands for q = 2^(31 - SB) + 1 + i -- above every key's, all different.
template <int G, int E>
__device__ __forceinline__ void sort_key_lists(unsigned long long *keys, int pi, int gq, int cnt, float r2) {
    static_assert(G == 4 || G == 8, "the lanes of a group are a DPP quad, or two quads of a row");
    constexpr int kPts = kLanes / G;
    constexpr int SB = G * E == 64 ? 6 : G * E == 128 ? 7 : 8;
    static_assert((1 << SB) == G * E, "slot bits");
    constexpr unsigned kQEnd = 1u << (31 - SB);
    const float scale = pin_f((float)kQEnd * 0.999f / r2);           // q < kQEnd for every d2 < r2
    wave_lds_fence();
    if (scale < 3.0e38f) {                                           // (uniform; a radius so small that the scale overflows: below)
        unsigned r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned idx = (unsigned)(gq * E + e);
            const float d2 = __uint_as_float(reinterpret_cast<const unsigned *>(keys)[2 * (min((int)idx, cnt - 1 < 0 ? 0 : cnt - 1) * kPts + pi) + 1]);
            const unsigned q = (unsigned)(d2 * scale);
            r[e] = (((int)idx < cnt ? q : kQEnd + 1u + idx) << SB) | idx;
        }
        KeySort32<G, E>::template level<2>(r, gq);
        unsigned near = ~0u;                                         // smallest difference pattern of two neighbours
#pragma unroll
        for (int e = 0; e + 1 < E; ++e) near = min(near, r[e] ^ r[e + 1]);
        const unsigned nxt = (unsigned)__shfl_down((int)r[0], 1);
        if (gq != G - 1) near = min(near, r[E - 1] ^ nxt);
        if (!__any((near >> SB) == 0u)) {
            unsigned long long k64[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (gq * E + e < cnt) k64[e] = keys[(int)(r[e] & (unsigned)(G * E - 1)) * kPts + pi];
                if (e % 8 == 7) __builtin_amdgcn_sched_barrier(0);      // (a stand-in's register is free once its key is requested)
            }
            wave_lds_fence();                                        // every lane has read its keys
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (gq * E + e < cnt) keys[(gq * E + e) * kPts + pi] = k64[e];
            wave_lds_fence();
            return;
        }
    }
    unsigned long long r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = gq * E + e;
        r[e] = idx < cnt ? keys[idx * kPts + pi] : ~0ull;
    }
    KeySort<G, E>::template level<2>(r, gq);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int idx = gq * E + e;
        if (idx < cnt) keys[idx * kPts + pi] = r[e];
    }
    wave_lds_fence();
}

constexpr int kSortedListKeys = 32 * kSortGroup;      // keys per point and pass: what the register sort holds

// keeps the keys below `hi` of every point's list, in place and in order; returns the new length.  lcap = list capacity
template <int G>
__device__ __forceinline__ int keep_keys_below(unsigned long long *keys, int pi, int gq, unsigned group_shift, int cnt, int lcap,
                                               unsigned long long hi) {
    constexpr int kPts = kLanes / G;
    int out = 0;
    wave_lds_fence();
    for (int base = 0; base < lcap; base += G) {
        const int idx = base + gq;
        const unsigned long long key = keys[min(idx, lcap - 1) * kPts + pi];
        const bool keep = (idx < cnt) & (key < hi);
        const unsigned gb = (unsigned)(__ballot(keep) >> group_shift) & ((1u << G) - 1u);
        wave_lds_fence();                                   // every lane has read its key: the writes land at or before them
        if (keep) keys[(out + __popc(gb & ((1u << gq) - 1u))) * kPts + pi] = key;
        out += __popc(gb);
        wave_lds_fence();
    }
    return out;
}

// (the sparse query kernel: no wave / workgroup kernels behind it -- a list that runs full is cut into windows)
template <int G>
__device__ __forceinline__ int point_features_sorted(const float4 *__restrict__ pts, const char *__restrict__ nrmsrc,
                                                     unsigned ns, const int *__restrict__ cell_start, const GridDesc &g,
                                                     const FeatDesc &fin, float4 p, float4 np, float *H, uint2 *ent, int ecap,
                                                     unsigned long long *keys, int lcap, bool active) {
    constexpr int kPts = kLanes / G;
    const int tid = threadIdx.x, pi = tid / G, gq = tid % G;
    FeatDesc f;
    f.A = pin_i(fin.A);
    f.B = pin_i(fin.B);
    f.F = pin_i(fin.F);
    f.A1f = pin_f(fin.A1f);
    f.B1f = pin_f(fin.B1f);
    f.support = fin.support;
    f.ann_dim = pin_f(fin.ann_dim);
    f.ann_half = pin_f(fin.ann_half);
    f.ann_rdim = pin_f(fin.ann_rdim);
    f.bin_dim = pin_f(fin.bin_dim);
    f.bin_half = pin_f(fin.bin_half);
    f.bin_rdim = pin_f(fin.bin_rdim);
    f.r2 = pin_f(fin.r2);
    f.rr = fin.rr;
    for (int c = gq; c < f.F; c += G) H[c * kPts + pi] = 0.0f;                     // hpp:325
    RowSearch<G> rs;
    rs.init(pts, cell_start, g, p, f.rr, f.r2, active, gq);
    const int ent_last = (ecap - 1) * kPts + pi;
    const int col_address = lds_address(H + pi);
    const unsigned group_shift = (unsigned)(tid & ~(G - 1));
    const unsigned long long key_end = (unsigned long long)__float_as_uint(f.r2) << 32;     // every key is below it (d2 < r2)
    // A pass collects the keys of the window [lo, hi) of the point, sorts and adds them.  hi starts open (key_end); when
    // a list is about to run full, hi is lowered to a pivot inside the window and the list is filtered -- the pass then
    // still ends with ALL keys of [lo, hi), and the next pass takes [hi, key_end).
    unsigned long long lo = 0ull, hi = key_end;
    bool dropped = false;                          // hpp:336: element 0 of the whole order has been dropped
    int kf = 0;
    for (int pass = 0;; ++pass) {
        int cnt = 0;                               // keys in the point's list (the same in the lanes of the group)
        rs.restart();
        for (;;) {
            // ---- search: accept words of the point
            int ecnt = 0;
            bool full = false;
            while (!full) {
                int wbase;
                unsigned w;
                if (!rs.next_word(wbase, w)) break;
                if (pass == 0) kf += __popc(w);
                if (w != 0u) {
                    if (gq == 0) ent[ecnt * kPts + pi] = make_uint2((unsigned)wbase, __brev(w));     // first candidate = bit 0
                    ++ecnt;
                }
                full = __any(ecnt == ecap);
            }
            // ---- collect: G accepted neighbors of the point per round -> keys
            if (__any(ecnt > 0)) {
                wave_lds_fence();
                int e = 0;
                unsigned w = 0u;
                int wbase = 0;
                uint2 nw = ent[pi];
                struct Taken {
                    bool valid;
                    float4 q;
                };
                auto take = [&](Taken &slot) {
                    const bool refill = (w == 0u) & (e < ecnt);
                    w = refill ? nw.y : w;
                    wbase = refill ? (int)nw.x : wbase;
                    e += refill ? 1 : 0;
                    nw = ent[min(e * kPts + pi, ent_last)];
                    const unsigned c1 = drop_lowest_bit(w), c2 = drop_lowest_bit(c1), c3 = drop_lowest_bit(c2);
                    const unsigned m = gq == 0 ? w : gq == 1 ? c1 : gq == 2 ? c2 : c3;
                    w = drop_lowest_bit(c3);
                    slot.valid = m != 0u;
                    const int tt = slot.valid ? wbase + lowest_bit_index(m) : 0;
                    slot.q = pts[tt];
                };
                auto collect = [&](Taken &now) {
                    // room for the G keys of this round in every list, else: a pivot inside the window, the list filtered
                    while (__any(cnt > lcap - G)) {
                        if (cnt > lcap - G) {
                            const unsigned lb = (unsigned)(lo >> 32), hb = (unsigned)(hi >> 32);
                            if (hb - lb >= 2u) {           // halve the window in d2 (its values, not its bits: even counts on a surface)
                                const float ld = __uint_as_float(lb), hd = __uint_as_float(hb);
                                unsigned mb = __float_as_uint(ld + (hd - ld) * 0.5f);
                                if (mb <= lb || mb >= hb) mb = lb + (hb - lb) / 2u;
                                hi = (unsigned long long)mb << 32;
                            } else {                       // one or two distances left: halve by the whole key (distance, index)
                                hi = lo + (hi - lo) / 2ull;
                            }
                        }
                        cnt = keep_keys_below<G>(keys, pi, gq, group_shift, cnt, lcap, hi);
                    }
                    const unsigned long long key = ((unsigned long long)__float_as_uint(dist2(p.x, p.y, p.z, now.q)) << 32) |
                                                   (unsigned long long)(unsigned)__float_as_int(now.q.w);
                    const bool app = now.valid & (key >= lo) & (key < hi);
                    const unsigned gb = (unsigned)(__ballot(app) >> group_shift) & ((1u << G) - 1u);
                    if (app) keys[(cnt + __popc(gb & ((1u << gq) - 1u))) * kPts + pi] = key;
                    cnt += __popc(gb);
                    now.valid = false;
                };
                Taken pa, pb;
                pa.valid = pb.valid = false;
                pa.q = pb.q = make_float4(0.f, 0.f, 0.f, 0.f);
                do {
                    take(pb);
                    collect(pa);
                    take(pa);
                    collect(pb);
                } while (__any((w != 0u) | (e < ecnt) | pa.valid | pb.valid));
            }
            if (rs.exhausted()) break;
        }
        // ---- sort the lists (the one place where the network is instantiated)
        if (lcap <= kSortedListKeys / 2) sort_key_lists<G, kSortedListKeys / G / 2>(keys, pi, gq, cnt, f.r2);     // (uniform)
        else sort_key_lists<G, kSortedListKeys / G>(keys, pi, gq, cnt, f.r2);
        // ---- add the neighbors in order, G per round; hpp:336: element 0 of the whole order is dropped
        {
            struct Next {
                bool valid;
                float d2;
                f32x3 n;
            };
            int k = (!dropped & (cnt > 0)) ? 1 : 0;
            dropped |= cnt > 0;
            const int key_last = (lcap - 1) * kPts + pi;
            auto take = [&](Next &slot) {
                const int idx = k + gq;
                slot.valid = idx < cnt;
                const unsigned long long key = keys[min(idx * kPts + pi, key_last)];
                k += G;
                slot.d2 = __uint_as_float((unsigned)(key >> 32));
                const unsigned orig = slot.valid ? (unsigned)key : 0u;
                slot.n = *reinterpret_cast<const f32x3 *>(nrmsrc + (size_t)orig * ns);
            };
            Next pa, pb;
            pa.valid = pb.valid = false;
            pa.d2 = pb.d2 = 0.f;
            pa.n = pb.n = f32x3{0.f, 0.f, 0.f};
#define KPL_SORTED_ROUND(now, nxt)                                                                 \
    {                                                                                              \
        take(nxt);                                                                                 \
        const bool has_ = now.valid & finite3(now.n.x, now.n.y, now.n.z);          /* hpp:338 */  \
        Contribution c_;                                                                           \
        if (has_) c_ = neighbor_contribution<kPts>(f, now.d2, np, now.n, col_address);             \
        _Pragma("unroll") for (int sub_ = 0; sub_ < G; ++sub_) {                                   \
            if (has_ & (gq == sub_)) apply_contribution(c_, request_cells(c_));                    \
            wave_lds_fence();                                                                      \
        }                                                                                          \
        now.valid = false;                                                                         \
    }
            do {
                KPL_SORTED_ROUND(pa, pb)
                KPL_SORTED_ROUND(pb, pa)
            } while (__any((k - G < cnt) | pa.valid | pb.valid));
#undef KPL_SORTED_ROUND
        }
        if (!__any(hi != key_end)) break;
        // the next pass: the keys from the pivot on; a point whose window was never cut is done (an empty window)
        lo = hi != key_end ? hi : key_end;
        hi = key_end;
        wave_lds_fence();
    }
    wave_lds_fence();
    for (int a = gq; a < f.A; a += G) {                                            // hpp:360-370, one row per lane
        float *h = H + (a * f.B) * kPts + pi;
        float s = 0.0f;
        for (int k = 0; k < f.B; ++k) {
            float v = h[k * kPts];
            s += v * v;
        }
        const float nr = sqrtf(s);
        if (nr > 0)
            for (int k = 0; k < f.B; ++k) h[k * kPts] = h[k * kPts] / nr;
    }
    wave_lds_fence();
    return kf;
}


// ---------------------------------------------------------------------------------------------
// Sorted order through the word lists ("sorted words": FeatDesc::sorted && walk == 1), for views whose points hold more
// neighbors than the register lists of point_features_sorted (124) and up to ~250: round 5 gave such a point a whole wave
// (sorted_collect_wave_kernel: its own search of the box, a 256 .. 512-key network, the keys through HBM to sorted_add_kernel
// -- ~1 100 + ~800 wave-instructions per point).  Here the search is the two-pass walk's (feature_search_kernel: the candidates
// of a wave's points staged in LDS once, accept words to a list in global memory; no neighbor dropped), and EIGHT lanes per
// point -- 8 points per wave -- do the rest in one kernel.  The kernel is a chain of dependent steps per wave (the ordered
// updates), so its throughput is its occupancy: 8 / 5 / 4 / 2 waves per CU took 4.2 / 6.8 / 8.5 / 16.5 ms (profiles/
// r06_notes.md).  Its LDS therefore holds FOUR bytes per neighbor -- the neighbor's storage position -- not the 8-byte key:
//   collect  lane g expands the words g, g + 8, ... of the point's list into positions (slot = neighbors before it)
//   sort     every lane loads the records of its 32 slots (12 bytes: xyz), d2 as everywhere -> the 32-bit stand-in
//            q(d2) << 8 | slot (see sort_key_lists); the network KeySort32 over 8 lanes x 32 stand-ins in registers; the
//            positions written back in that order
//   add      G positions per round: the neighbor's records in storage order -- xyz -> d2 again (the same arithmetic, the same
//            bits), its normal --, then as point_features_sorted (hpp:334-359; element 0 dropped, hpp:336)
// A point with more neighbors than the list holds, or with two neighbours in the sorted stand-ins that share a q (equal or
// almost equal distances: only the 64-bit keys (d2, index) order those), is listed for the wave / workgroup kernels like the
// deferred points of feature_sorted_kernel.  Same keys, same order, same arithmetic: the same bits.
//   LDS: [H: maxF x 8 floats][position lists: kWordsKeys x 8 positions of 4 bytes]
constexpr int kWordsGroup = 8, kWordsKeys = 32 * kWordsGroup;

// the positions of every point's list (tl[slot * kPts + point], cnt <= G E of them) into ascending (d2, index) order; returns
// true for a point whose order the stand-ins do not decide
template <int G, int E>
__device__ __forceinline__ bool sort_position_lists(unsigned *tl, int lcap, const float4 *__restrict__ pts, const float4 &p, int pi,
                                                    int gq, int cnt, float r2, unsigned group_shift) {
    constexpr int kPts = kLanes / G;
    constexpr int SB = G * E == 64 ? 6 : G * E == 128 ? 7 : G * E == 256 ? 8 : 9;
    static_assert((1 << SB) == G * E, "slot bits");
    constexpr unsigned kQEnd = 1u << (31 - SB);
    const float scale = pin_f((float)kQEnd * 0.999f / r2);
    if (!(scale < 3.0e38f)) return true;                             // (uniform) a radius so small that the scale overflows
    unsigned r[E];
    wave_lds_fence();
    const int last = max(cnt - 1, 0);                                // (slot 0 always holds a valid position)
#pragma unroll
    for (int e0 = 0; e0 < E; e0 += 8) {
        f32x3 c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = ld12(pts, (int)tl[min(gq * E + e0 + j, last) * kPts + pi]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned idx = (unsigned)(gq * E + e0 + j);
            const unsigned q = (unsigned)(dist2(p.x, p.y, p.z, c[j]) * scale);
            r[e0 + j] = (((int)idx < cnt ? q : kQEnd + 1u + idx) << SB) | idx;
        }
    }
    KeySort32<G, E>::template level<2>(r, gq);
    unsigned near = ~0u;
#pragma unroll
    for (int e = 0; e + 1 < E; ++e) near = min(near, r[e] ^ r[e + 1]);
    const unsigned nxt = (unsigned)__shfl_down((int)r[0], 1);
    if (gq != G - 1) near = min(near, r[E - 1] ^ nxt);
    const bool tie = (((unsigned)(__ballot((near >> SB) == 0u) >> group_shift)) & ((1u << G) - 1u)) != 0u;
    // (slots and positions from cnt on are never used; the list may be shorter than the network: lcap <= G E)
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = tl[min((int)(r[e] & (unsigned)(G * E - 1)), lcap - 1) * kPts + pi];
    wave_lds_fence();                                                // every lane has read its positions
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (gq * E + e < lcap) tl[(gq * E + e) * kPts + pi] = r[e];
    wave_lds_fence();
    return tie;
}

// the positions of the neighbors in the words 0 .. ecnt - 1 of a point (fetch(e): first position and bits of word e, zeros
// beyond the list) appended to its list: lane g of the group expands the words g, g + G, ... (slot = neighbors before it).
// cnt comes back as the number of neighbors, also where the list (lcap positions) does not hold them all
template <int G, class Fetch>
__device__ __forceinline__ void expand_words(unsigned *tl, int lcap, int pi, int gq, unsigned group_shift, int ecnt, int &cnt, Fetch fetch) {
    constexpr int kPts = kLanes / G;
    uint2 wd = fetch(gq);
    for (int eb = 0; __any(eb < ecnt); eb += G) {
        const uint2 nx = fetch(eb + G + gq);                                   // (the next G words on their way)
        unsigned bits = wd.y;
        int incl = __popc(bits);                                               // neighbors in the group's words up to this lane's
        {
            const int s1 = __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);          // row_shr:1
            incl += gq >= 1 ? s1 : 0;
            const int s2 = __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);          // row_shr:2
            incl += gq >= 2 ? s2 : 0;
            if (G == 8) {
                const int s4 = __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);      // row_shr:4
                incl += gq >= 4 ? s4 : 0;
            }
        }
        int slot = cnt + incl - __popc(bits);
        cnt += __shfl(incl, (int)group_shift + G - 1);
        while (__any(bits != 0u)) {
            if (bits != 0u) {
                if (slot < lcap) tl[slot * kPts + pi] = wd.x + (unsigned)lowest_bit_index(bits);
                ++slot;
                bits = drop_lowest_bit(bits);
            }
        }
        wd = nx;
    }
}

// the feature loop (hpp:334-359) over a point's positions in sorted order, G per round (element 0 of the order is
// dropped, hpp:336).  A position names the neighbor's records in STORAGE order: xyz -> d2 again (the arithmetic the stand-in
// saw: the same bits), and the normal as the canonical kernels read it (x = NaN where it is not finite: cell_sort_store) --
// two 12-byte reads next to those of the point's other neighbors, requested three rounds ahead, where the 64-bit keys
// (d2, original index) of the other kernels need a gather into the caller's normal array per neighbor.
template <int G>
__device__ __forceinline__ void add_sorted_positions(const unsigned *tl, int lcap, int cnt, const float4 *__restrict__ pts,
                                                     const float4 *__restrict__ nrm, const FeatDesc &f, const float4 &p,
                                                     const float4 &np, int col_address, int pi, int gq) {
    constexpr int kPts = kLanes / G;
    struct Next {
        bool valid;
        f32x3 q, n;
    };
    int k = cnt > 0 ? 1 : 0;
    auto request = [&](Next &s) {
        const int idx = k + gq;
        s.valid = idx < cnt;
        const unsigned t = tl[min(idx, lcap - 1) * kPts + pi];
        k += G;
        const int tt = s.valid ? (int)t : 0;
        s.q = ld12(pts, tt);
        s.n = ld12(nrm, tt);
    };
    constexpr int kAhead = 3;
    Next sl[kAhead + 1];
#pragma unroll
    for (int q = 0; q <= kAhead; ++q) {
        sl[q].valid = false;
        sl[q].q = sl[q].n = f32x3{0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < kAhead; ++q) request(sl[q]);
    bool more = true;
    while (more) {
#pragma unroll
        for (int q = 0; q <= kAhead; ++q) {
            Next &now = sl[q];
            request(sl[(q + kAhead) % (kAhead + 1)]);
            const bool has_ = now.valid & (now.n.x == now.n.x);                        /* hpp:338 */
            Contribution c_;
            if (has_) c_ = neighbor_contribution<kPts>(f, dist2(p.x, p.y, p.z, now.q), np, now.n, col_address);
#pragma unroll
            for (int sub_ = 0; sub_ < G; ++sub_) {
                if (has_ & (gq == sub_)) apply_contribution(c_, request_cells(c_));
                wave_lds_fence();
            }
            now.valid = false;
        }
        bool pending = false;
#pragma unroll
        for (int q = 0; q <= kAhead; ++q) pending |= sl[q].valid;
        more = __any(pending);
    }
}

// hpp:360-370: the rows of the histogram normalized, one row per lane of the group
template <int G>
__device__ __forceinline__ void normalize_rows(float *H, const FeatDesc &f, int pi, int gq) {
    constexpr int kPts = kLanes / G;
    wave_lds_fence();
    for (int a = gq; a < f.A; a += G) {
        float *h = H + (a * f.B) * kPts + pi;
        float sq = 0.0f;
        for (int kk = 0; kk < f.B; ++kk) {
            float v = h[kk * kPts];
            sq += v * v;
        }
        const float nr = sqrtf(sq);
        if (nr > 0)
            for (int kk = 0; kk < f.B; ++kk) h[kk * kPts] = h[kk * kPts] / nr;
    }
    wave_lds_fence();
}

__device__ __forceinline__ FeatDesc pinned_feat(const FeatDesc &fin) {
    FeatDesc f;
    f.A = pin_i(fin.A);
    f.B = pin_i(fin.B);
    f.F = pin_i(fin.F);
    f.A1f = pin_f(fin.A1f);
    f.B1f = pin_f(fin.B1f);
    f.support = fin.support;
    f.ann_dim = pin_f(fin.ann_dim);
    f.ann_half = pin_f(fin.ann_half);
    f.ann_rdim = pin_f(fin.ann_rdim);
    f.bin_dim = pin_f(fin.bin_dim);
    f.bin_half = pin_f(fin.bin_half);
    f.bin_rdim = pin_f(fin.bin_rdim);
    f.r2 = pin_f(fin.r2);
    f.rr = fin.rr;
    return f;
}

template <int G, int EMAX>
__device__ __forceinline__ int point_features_sorted_words(const float4 *__restrict__ pts, const float4 *__restrict__ nrm,
                                                           const FeatDesc &fin, float4 p, float4 np, float *H,
                                                           unsigned *tl, const uint2 *__restrict__ list, int stride,
                                                           int ecnt, bool &deferred) {
    static_assert(G == 8 && (EMAX == 32 || EMAX == 64), "eight lanes per point: up to 256 neighbors, or 512");
    constexpr int kPts = kLanes / G, lcap = EMAX * G;
    const int tid = threadIdx.x, pi = tid / G, gq = tid % G;
    const FeatDesc f = pinned_feat(fin);
    for (int c = gq; c < f.F; c += G) H[c * kPts + pi] = 0.0f;                     // hpp:325
    if (gq == 0) tl[pi] = 0u;
    const int col_address = lds_address(H + pi);
    const unsigned group_shift = (unsigned)(tid & ~(G - 1));
    deferred = false;
    int cnt = 0;
    // ---- collect: the words of the list, G at a time, each expanded by its lane
    expand_words<G>(tl, lcap, pi, gq, group_shift, ecnt, cnt,
                    [&](int e) { return e < ecnt ? list[e * stride] : make_uint2(0u, 0u); });
    const int kf = cnt;
    if (cnt > lcap) {                  // more neighbors than the list holds: the point leaves for the wave / workgroup kernels
        deferred = true;
        cnt = 0;
    }
    // ---- sort: 32 G stand-ins, or half of that when no list of the wave holds more
    {
        bool tie;
        if (__all(cnt <= 16 * G)) tie = sort_position_lists<G, 16>(tl, lcap, pts, p, pi, gq, cnt, f.r2, group_shift);
        else if (EMAX == 32 || __all(cnt <= 32 * G)) tie = sort_position_lists<G, 32>(tl, lcap, pts, p, pi, gq, cnt, f.r2, group_shift);
        else tie = sort_position_lists<G, EMAX>(tl, lcap, pts, p, pi, gq, cnt, f.r2, group_shift);
        if (tie & (cnt > 1)) {         // equal or almost equal distances: the 64-bit keys order them (the wave kernel)
            deferred = true;
            cnt = 0;
        }
    }
    add_sorted_positions<G>(tl, lcap, cnt, pts, nrm, f, p, np, col_address, pi, gq);
    normalize_rows<G>(H, f, pi, gq);
    return kf;
}

// The whole-view kernel of the sorted order (feature_sorted_kernel: up to 124 neighbors per point, kSortGroup lanes per point):
// the search of point_features_sorted (RowSearch: accept words of the point's box, `ecap` of them in LDS at a time), then as
// point_features_sorted_words -- positions, stand-ins, ordered adds; 4 bytes of LDS per neighbor and ~90 registers keep
// ~16 waves per CU where the 64-bit key lists kept 8-10.  overflow: the list ran full (the search stops: kf is not the
// neighborhood's size then); deferred: overflow, or an order the stand-ins do not decide -- a point for the wave / workgroup kernels.
template <int G>
__device__ __forceinline__ int point_features_sorted_view(const float4 *__restrict__ pts, const float4 *__restrict__ nrm,
                                                          const int *__restrict__ cell_start, const GridDesc &g, const FeatDesc &fin,
                                                          float4 p, float4 np, float *H, uint2 *ent, int ecap, unsigned *tl, int lcap,
                                                          bool active, bool &deferred, bool &overflow) {
    constexpr int kPts = kLanes / G;
    const int tid = threadIdx.x, pi = tid / G, gq = tid % G;
    const FeatDesc f = pinned_feat(fin);
    for (int c = gq; c < f.F; c += G) H[c * kPts + pi] = 0.0f;                     // hpp:325
    if (gq == 0) tl[pi] = 0u;
    RowSearch<G> rs;
    rs.init(pts, cell_start, g, p, f.rr, f.r2, active, gq);
    const int col_address = lds_address(H + pi);
    const unsigned group_shift = (unsigned)(tid & ~(G - 1));
    deferred = overflow = false;
    int cnt = 0, kf = 0;
    rs.restart();
    for (;;) {
        // ---- search: accept words of the point
        int ecnt = 0;
        bool full = false;
        while (!full) {
            int wbase;
            unsigned w;
            if (!rs.next_word(wbase, w)) break;
            kf += __popc(w);
            if (w != 0u) {
                if (gq == 0) ent[ecnt * kPts + pi] = make_uint2((unsigned)wbase, __brev(w));     // first candidate = bit 0
                ++ecnt;
            }
            full = __any(ecnt == ecap);
        }
        // ---- collect: their neighbors' positions
        if (__any(ecnt > 0)) {
            wave_lds_fence();
            int c2 = cnt;
            expand_words<G>(tl, lcap, pi, gq, group_shift, ecnt, c2,
                            [&](int e) { return e < ecnt ? ent[e * kPts + pi] : make_uint2(0u, 0u); });
            wave_lds_fence();
            if (!overflow) cnt = c2;
            if (cnt > lcap) {          // the list is full: the point leaves (its rows are over: it does not hold up the others' walk)
                overflow = true;
                cnt = 0;
                rs.ny = 0;
                rs.t1 = rs.t;
            }
        }
        if (rs.exhausted()) break;
        if (!__any(active && !overflow)) break;           // every point of the wave has left for the large path
    }
    deferred = overflow;
    {
        const bool tie = (lcap <= 16 * G || __all(cnt <= 16 * G)) ? sort_position_lists<G, 16>(tl, lcap, pts, p, pi, gq, cnt, f.r2, group_shift)
                                                                  : sort_position_lists<G, 32>(tl, lcap, pts, p, pi, gq, cnt, f.r2, group_shift);
        if (tie & (cnt > 1)) {
            deferred = true;
            cnt = 0;
        }
    }
    add_sorted_positions<G>(tl, lcap, cnt, pts, nrm, f, p, np, col_address, pi, gq);
    normalize_rows<G>(H, f, pi, gq);
    return kf;
}

// runForest, hpp:267-296 + cv::ml::RTrees::predict(PREDICT_SUM) restated (hpp:281): per tree
// walk "val <= thr ? left : right", double sum of leaf values in tree order, (float)sum,
// score = 1 - sum / (T * 1.0f).  Several trees are walked at once per lane so that several
// dependent node reads are in flight.  The walks are written without data-dependent branches: the
// split variable of a leaf reads feature 0 and is ignored, and all node reads of a level are issued
// before the first one is used, then all feature reads -- one LDS round trip each per level for all
// ways together.
//
// Where the nodes come from (layout: forest.h):
//   top part, slots < ntop   level-major; its first `nlds` slots are staged in LDS by the forest
//                            kernels (all of it unless the histograms leave less than 64 KB); the
//                            rest, if any, is read node by node from global memory
//   blocks, slots >= ntop    8-slot (64-byte) subtrees of three levels: a lane fetches the whole block
//                            with four 16-byte loads and walks its three levels out of registers.
//                            Node by node, 64 lanes at 64 different nodes are 64 cache lines per level,
//                            and with thousands of such lines in flight per CU a line does not survive
//                            in the L1 until the walk comes back for the next level: the deep walk of
//                            config 5 moved 28 GB from the L2 for 1.8 GB of nodes (profiles/r02_notes.md).
constexpr int kTreeWays = 10;    // forest entirely in LDS
constexpr int kChainWays = 4;    // chained forest (forest_sum_chained): walks per lane
constexpr int kDeepWays = 4;     // forest with blocks: 16 VGPRs of block per way

template <int WAYS>
__device__ __forceinline__ void fetch_nodes(const ForestDev &forest, const uint2 *lnodes, uint32_t last_lds,
                                            bool all_in_lds, const uint32_t (&nd)[WAYS], uint2 (&node)[WAYS]) {
#pragma unroll
    for (int k = 0; k < WAYS; ++k) node[k] = lnodes[all_in_lds ? nd[k] : min(nd[k], last_lds)];
    if (!all_in_lds) {      // nodes beyond the LDS: global memory, all ways issued before the first use
        bool far_any = false;
#pragma unroll
        for (int k = 0; k < WAYS; ++k) far_any |= nd[k] > last_lds;
        if (__any(far_any)) {
            uint2 far[WAYS];
#pragma unroll
            for (int k = 0; k < WAYS; ++k)      // 32-bit byte offset from the uniform base (< 2^24 slots of 8 bytes)
                far[k] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(forest.nodes) +
                                                          ((nd[k] > last_lds ? nd[k] : 0u) << 3));
#pragma unroll
            for (int k = 0; k < WAYS; ++k) node[k] = nd[k] > last_lds ? far[k] : node[k];
        }
    }
}

// The whole forest is in LDS (the leanest loop): kTreeWays trees in step, a finished walk keeps re-reading
// its leaf until the others end.
template <bool STATS>
__device__ __forceinline__ float forest_sum(const ForestDev &forest, const uint2 *lnodes, int nlds,
                                            const float *x, int nvars, int &depth) {
    double sum = 0.0;
    const uint32_t last_lds = (uint32_t)(nlds - 1), last_var = (uint32_t)nvars - 1u;
    for (int t0 = 0; t0 < forest.ntrees; t0 += kTreeWays) {
        uint32_t nd[kTreeWays];
        uint2 node[kTreeWays];
        bool done[kTreeWays];          // STATS: the walk had reached its leaf before this read
#pragma unroll
        for (int k = 0; k < kTreeWays; ++k) {
            nd[k] = t0 + k < forest.ntrees ? t0 + k : t0;       // level-major layout: the root of tree t is node t
            done[k] = t0 + k >= forest.ntrees;
        }
        for (;;) {
            fetch_nodes<kTreeWays>(forest, lnodes, last_lds, true, nd, node);
            float val[kTreeWays];
            bool leaf[kTreeWays];
#pragma unroll
            for (int k = 0; k < kTreeWays; ++k) {
                const uint32_t var = node[k].y >> 24;
                leaf[k] = var == 255u;
                val[k] = x[min(var, last_var) * kLanes];        // a leaf reads the last feature and ignores it
            }
            bool all_done = true;
#pragma unroll
            for (int k = 0; k < kTreeWays; ++k) {
                const uint32_t next = (node[k].y & 0x00ffffffu) + (val[k] <= __uint_as_float(node[k].x) ? 0u : 1u);
                if (STATS) depth += done[k] ? 0 : 1;            // visited nodes: internal ones and the leaf, once
                if (STATS) done[k] = leaf[k];
                nd[k] = leaf[k] ? nd[k] : next;
                all_done &= leaf[k];
            }
            if (__all(all_done)) break;
        }
#pragma unroll
        for (int k = 0; k < kTreeWays; ++k)
            if (t0 + k < forest.ntrees) sum += (double)__uint_as_float(node[k].x);
    }
    return (float)sum;
}

// The same in-step walk for a lane that takes every tstride-th tree starting at `first` (a point shared by tstride
// lanes): at most WAYS trees per lane, the whole forest in LDS, features x[var * xstride].  Returns the lane's part of
// the sum as a float -- exact, and independent of the split, when the leaf values are small integers (order_free).
template <bool STATS, int WAYS>
__device__ __forceinline__ float forest_sum_strided(const ForestDev &forest, const uint2 *lnodes, const float *x, int xstride,
                                                    int nvars, int first, int tstride, int &depth) {
    const uint32_t last_var = (uint32_t)nvars - 1u;
    uint32_t nd[WAYS];
    uint2 node[WAYS];
    bool mine[WAYS], done[WAYS];
#pragma unroll
    for (int k = 0; k < WAYS; ++k) {
        mine[k] = first + tstride * k < forest.ntrees;
        nd[k] = mine[k] ? (uint32_t)(first + tstride * k) : (uint32_t)forest.ntrees;     // no tree: the resting leaf (value 0)
        done[k] = !mine[k];
    }
    for (;;) {
#pragma unroll
        for (int k = 0; k < WAYS; ++k) node[k] = lnodes[nd[k]];
        float val[WAYS];
        bool leaf[WAYS];
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            const uint32_t var = node[k].y >> 24;
            leaf[k] = var == 255u;
            val[k] = x[__umul24(min(var, last_var), (uint32_t)xstride)];      // a leaf reads the last feature and ignores it
        }
        bool all_done = true;
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            const uint32_t next = (node[k].y & 0x00ffffffu) + (val[k] <= __uint_as_float(node[k].x) ? 0u : 1u);
            if (STATS) depth += done[k] ? 0 : 1;
            if (STATS) done[k] = leaf[k];
            nd[k] = leaf[k] ? nd[k] : next;
            all_done &= leaf[k];
        }
        if (__all(all_done)) break;
    }
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < WAYS; ++k) sum += __uint_as_float(node[k].x);           // (the resting leaf adds 0)
    return sum;
}

// WAYS trees of a forest with blocks, walked to their leaves: tree[k] < 0 = no tree (leaf value 0).
//   phase 1  the top part, one level per step, until every walk is at a leaf or at the root of a block
//   phase 2  one block = up to three levels per step.  A finished walk fetches block 0 (one line shared
//            by every such lane: the requests of a wave are served line by line).
// The features of the lane's point are x[var * xstride].
template <bool STATS, int WAYS>
__device__ __forceinline__ void walk_deep(const ForestDev &forest, const uint2 *lnodes, int nlds, const float *x,
                                          int xstride, const int (&tree)[WAYS], float (&leafval)[WAYS], int &depth) {
    const uint32_t last_lds = (uint32_t)(nlds - 1), ntop = (uint32_t)forest.ntop;
    const bool top_in_lds = (uint32_t)nlds >= ntop;
    uint32_t nd[WAYS];
    bool done[WAYS];
#pragma unroll
    for (int k = 0; k < WAYS; ++k) {
        done[k] = tree[k] < 0;
        nd[k] = done[k] ? 0u : (uint32_t)tree[k];               // level-major layout: the root of tree t is node t
        leafval[k] = 0.0f;
    }
    for (;;) {                                                  // ---- phase 1
        bool go = false;
#pragma unroll
        for (int k = 0; k < WAYS; ++k) go |= !done[k] & (nd[k] < ntop);
        if (!__any(go)) break;
        uint32_t at[WAYS];
        uint2 node[WAYS];
#pragma unroll
        for (int k = 0; k < WAYS; ++k) at[k] = (done[k] | (nd[k] >= ntop)) ? 0u : nd[k];
        fetch_nodes<WAYS>(forest, lnodes, last_lds, top_in_lds, at, node);
        float val[WAYS];
        bool leaf[WAYS];
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            const uint32_t var = node[k].y >> 24;
            leaf[k] = var == 255u;
            val[k] = x[(leaf[k] ? 0u : var) * xstride];
        }
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            const bool walking = !done[k] & (nd[k] < ntop);
            const uint32_t child = node[k].y & 0x00ffffffu;
            // siblings are adjacent in the top part, their blocks 8 slots apart below it
            const uint32_t next = child + (val[k] <= __uint_as_float(node[k].x) ? 0u : (child >= ntop ? 8u : 1u));
            if (STATS) depth += walking ? 1 : 0;
            leafval[k] = (walking & leaf[k]) ? __uint_as_float(node[k].x) : leafval[k];
            done[k] |= walking & leaf[k];
            nd[k] = (walking & !leaf[k]) ? next : nd[k];
        }
    }
    for (;;) {                                                  // ---- phase 2
        bool go = false;
#pragma unroll
        for (int k = 0; k < WAYS; ++k) go |= !done[k];
        if (!__any(go)) break;
        uint4 blk[WAYS][4];                                     // slots (0 1) (2 3) (4 5) (6 -)
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            const char *src = reinterpret_cast<const char *>(forest.nodes) + ((done[k] ? 0u : nd[k]) << 3);
#pragma unroll
            for (int j = 0; j < 4; ++j) blk[k][j] = *reinterpret_cast<const uint4 *>(src + 16 * j);
        }
        // level by level for all ways, so that the feature reads of a level are in flight together
        uint2 n0[WAYS], n1[WAYS], n2[WAYS];
        float v[WAYS];
        bool r0[WAYS], r1[WAYS];
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            n0[k] = make_uint2(blk[k][0].x, blk[k][0].y);
            const uint32_t var = n0[k].y >> 24;
            v[k] = x[(var == 255u ? 0u : var) * xstride];
        }
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            r0[k] = !(v[k] <= __uint_as_float(n0[k].x));
            n1[k] = r0[k] ? make_uint2(blk[k][1].x, blk[k][1].y) : make_uint2(blk[k][0].z, blk[k][0].w);
            const uint32_t var = n1[k].y >> 24;
            v[k] = x[(var == 255u ? 0u : var) * xstride];
        }
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            r1[k] = !(v[k] <= __uint_as_float(n1[k].x));
            const uint2 a = r1[k] ? make_uint2(blk[k][2].x, blk[k][2].y) : make_uint2(blk[k][1].z, blk[k][1].w);   // slots 4 : 3
            const uint2 b = r1[k] ? make_uint2(blk[k][3].x, blk[k][3].y) : make_uint2(blk[k][2].z, blk[k][2].w);   // slots 6 : 5
            n2[k] = r0[k] ? b : a;
            const uint32_t var = n2[k].y >> 24;
            v[k] = x[(var == 255u ? 0u : var) * xstride];
        }
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            const bool l0 = (n0[k].y >> 24) == 255u, l1 = (n1[k].y >> 24) == 255u, l2 = (n2[k].y >> 24) == 255u;
            const bool ends = l0 | l1 | l2;
            const uint32_t lv = l0 ? n0[k].x : l1 ? n1[k].x : n2[k].x;
            if (STATS) depth += done[k] ? 0 : (l0 ? 1 : l1 ? 2 : 3);
            leafval[k] = (!done[k] & ends) ? __uint_as_float(lv) : leafval[k];
            nd[k] = (n2[k].y & 0x00ffffffu) + (v[k] <= __uint_as_float(n2[k].x) ? 0u : 8u);
            done[k] |= ends;
        }
    }
}

// Forest with blocks, one lane per point: kDeepWays trees at a time, leaf values added in tree order (hpp:281).
template <bool STATS>
__device__ __forceinline__ float forest_sum_deep(const ForestDev &forest, const uint2 *lnodes, int nlds,
                                                 const float *x, int &depth) {
    double sum = 0.0;
    for (int t0 = 0; t0 < forest.ntrees; t0 += kDeepWays) {
        int tree[kDeepWays];
        float leafval[kDeepWays];
#pragma unroll
        for (int k = 0; k < kDeepWays; ++k) tree[k] = t0 + k < forest.ntrees ? t0 + k : -1;
        walk_deep<STATS, kDeepWays>(forest, lnodes, nlds, x, kLanes, tree, leafval, depth);
#pragma unroll
        for (int k = 0; k < kDeepWays; ++k)
            if (t0 + k < forest.ntrees) sum += (double)leafval[k];
    }
    return (float)sum;
}

// Node reads of a forest that is only partly in LDS, without merging two paths in registers: per walk ONE register
// pair, written by an LDS read under the exec mask of the lanes whose node is staged and by a global load under the mask
// of the others (byte offset slot x 8 in both address spaces: `lnodes` is at LDS address 0).  The two never write the
// same lane, so neither has to wait for the other, and a walk none of whose lanes is beyond the staged part issues no
// load instruction at all.  All reads of the four walks are in flight before the one wait.  (The compiler cannot express
// this: it would wait for the LDS data before it lets the load overwrite the register, or select between two results --
// 8 instructions per walk and step where this takes 5.)
#define KPL_FETCH_ONE(K)                                                                                           \
    "s_andn2_b64 exec, %[sv], %[f" #K "]\n\t"                                                                      \
    "ds_read_b64 %[n" #K "], %[a" #K "]\n\t"                                                                       \
    "s_and_b64 exec, %[sv], %[f" #K "]\n\t"                                                                        \
    "s_cbranch_scc0 " #K "f\n\t"                                                                                   \
    "global_load_dwordx2 %[n" #K "], %[a" #K "], %[base]\n\t"                                                      \
    #K ":\n\t"
__device__ __forceinline__ void fetch_nodes_masked(const uint2 *gnodes, uint32_t last_lds_byte, const uint32_t (&at)[4],
                                                   uint2 (&node)[4]) {
    unsigned long long r0, r1, r2, r3, f0, f1, f2, f3, saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "v_cmp_lt_u32 %[f0], %[last], %[a0]\n\t"          // lanes whose node is beyond the staged part
                 "v_cmp_lt_u32 %[f1], %[last], %[a1]\n\t"          // (all four compared while every lane is enabled)
                 "v_cmp_lt_u32 %[f2], %[last], %[a2]\n\t"
                 "v_cmp_lt_u32 %[f3], %[last], %[a3]\n\t"
                 KPL_FETCH_ONE(0) KPL_FETCH_ONE(1) KPL_FETCH_ONE(2) KPL_FETCH_ONE(3)
                 "s_mov_b64 exec, %[sv]\n\t"
                 "s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : [n0] "=&v"(r0), [n1] "=&v"(r1), [n2] "=&v"(r2), [n3] "=&v"(r3), [sv] "=&s"(saved),
                   [f0] "=&s"(f0), [f1] "=&s"(f1), [f2] "=&s"(f2), [f3] "=&s"(f3)
                 : [last] "s"(last_lds_byte), [a0] "v"(at[0]), [a1] "v"(at[1]), [a2] "v"(at[2]), [a3] "v"(at[3]), [base] "s"(gnodes)
                 : "scc");
    node[0] = make_uint2((uint32_t)r0, (uint32_t)(r0 >> 32));
    node[1] = make_uint2((uint32_t)r1, (uint32_t)(r1 >> 32));
    node[2] = make_uint2((uint32_t)r2, (uint32_t)(r2 >> 32));
    node[3] = make_uint2((uint32_t)r3, (uint32_t)(r3 >> 32));
}
#undef KPL_FETCH_ONE

// A forest whose leaf values are small integers (class labels: every forest the reference trains,
// src/main_train_detector.cpp:405-407): the double sum of hpp:281 is then exact in any order and equals an
// int32 sum, so the trees of a point may be shared out over several lanes and need not be kept in step.
// The lane walks the trees first, first + tstride, ...; its features are x[var * xstride].  Each of its
// kQueueWays walks takes the lane's next tree as soon as it reaches a leaf; the loop runs for about
// sum-of-depths / ways steps instead of (trees / ways) x the depth of the deepest tree.
//
// A walk without a tree sits on the forest's resting leaf (slot `ntrees`, value 0, forest.h): it "reaches a
// leaf" at every step, adds 0 and stays -- no per-walk flags, the loop is bound by VALU issue.
//
// (The whole forest is in LDS here; forests that are not take forest_sum_chained or forest_sum_deep.)
template <bool STATS, int kQueueWays>
__device__ __forceinline__ int forest_sum_any_order(const ForestDev &forest, const uint2 *lnodes, int nlds,
                                                    const float *x, int xstride, int nvars, int first, int tstride,
                                                    bool active, int &depth) {
    const uint32_t last_lds = (uint32_t)(nlds - 1);
    const uint32_t rest = (uint32_t)forest.ntrees;                  // the resting leaf
    const uint32_t last_var = (uint32_t)nvars - 1u;
    const int ntrees = active ? forest.ntrees : 0;
    // the leaf values are small integers (order_free): their float sum is exact in any order (|sum| < 2^24) and needs no
    // conversion per leaf; the resting leaf is slot `ntrees`, so "the lane's next tree, or rest" is min(next_tree, ntrees)
    float sum = 0.0f;
    // per lane: its first kQueueWays trees are taken; a lane without a point never takes one (its queue is "past the end")
    int next_tree = active ? first + tstride * kQueueWays : 0x3fffffff;
    uint32_t nd[kQueueWays];
#pragma unroll
    for (int k = 0; k < kQueueWays; ++k)               // level-major layout: the root of tree t is node t
        nd[k] = first + tstride * k < ntrees ? (uint32_t)(first + tstride * k) : rest;
    for (;;) {
        uint2 node[kQueueWays];
        fetch_nodes<kQueueWays>(forest, lnodes, last_lds, true, nd, node);
        float val[kQueueWays];
#pragma unroll
        for (int k = 0; k < kQueueWays; ++k)           // a leaf (var = 255) reads the last feature and ignores it
            val[k] = x[__umul24(min(node[k].y >> 24, last_var), (uint32_t)xstride)];
        bool walking = false;
#pragma unroll
        for (int k = 0; k < kQueueWays; ++k) {
            const bool leaf = (node[k].y >> 24) == 255u;
            const uint32_t child = node[k].y & 0x00ffffffu;
            const uint32_t next = child + (val[k] <= __uint_as_float(node[k].x) ? 0u : 1u);
            if (STATS) depth += nd[k] != rest ? 1 : 0;
            sum += leaf ? __uint_as_float(node[k].x) : 0.0f;
            // a walk that has reached its leaf takes the lane's next tree (its root is node next_tree) or rests
            nd[k] = leaf ? min((uint32_t)next_tree, rest) : next;
            next_tree += leaf ? tstride : 0;           // (runs on past ntrees while the lane rests: a few steps)
            walking |= nd[k] != rest;
        }
        if (!__any(walking)) break;
    }
    return (int)sum;
}

// The any-order sum (see forest_sum_any_order) with the tree queue taken out of the loop: the leaf records of a chained
// forest (forest.h: level-major throughout, siblings adjacent) link the trees t, t + chain, t + 2 chain, ..., the last one
// to the resting leaf, which points at itself.  Walk k of the lane starts at the root of tree first + tstride * k and
// just follows the records (chain = tstride x kChainWays); the loop ends when every walk of the wave rests.
//   * a walk is kept as the BYTE offset of its node (slot x 8): the address of both node reads
//   * a walk that rests is at a leaf: the "is every walk resting" test is run only when every walk of every lane is at
//     one, and the wave looks every second step
// A wave issues its instructions one after the other and four waves per SIMD hide little of it: what counts here is
// the NUMBER of instructions per walk and step, of any kind -- 19 where the tree queue of forest_sum_any_order with the
// two node paths merged in registers took 45 (config 5: forest kernel 1.23 -> 0.88 ms; 75 instructions per step of
// the wave, one per 5 cycles of the SIMD).  The price is that a walk cannot take over trees from a slower one of the
// same lane (config 5: 91 steps per 16 points against 82; 49 if no walk ever idled -- profiles/r03_notes.md).
template <bool STATS, bool DEEP>
__device__ __forceinline__ int forest_sum_chained(const ForestDev &forest, const uint2 *lnodes, int nlds,
                                                  const float *x, int xstride, int nvars, int first, int tstride,
                                                  bool active, int &depth) {
    constexpr int WAYS = kChainWays;
    static_assert(WAYS == 4, "fetch_nodes_masked is written for four walks");
    const uint32_t last_lds_byte = (uint32_t)(nlds - 1) << 3;
    const uint32_t rest = (uint32_t)forest.ntrees << 3;             // the resting leaf
    const uint32_t last_var = (uint32_t)nvars - 1u;
    const int ntrees = active ? forest.ntrees : 0;
    const uint32_t xbase = (uint32_t)lds_address(x), xbytes = (uint32_t)xstride * 4u;
    float sum = 0.0f;
    uint32_t at[WAYS];
#pragma unroll
    for (int k = 0; k < WAYS; ++k)                     // level-major layout: the root of tree t is node t
        at[k] = first + tstride * k < ntrees ? (uint32_t)(first + tstride * k) << 3 : rest;
    const unsigned long long every_lane = __ballot(true);
    auto step = [&]() -> unsigned long long {              // one step of the four walks; returns the lanes whose walks are ALL at a leaf
        uint2 node[WAYS];
        if (DEEP) {
            fetch_nodes_masked(forest.nodes, last_lds_byte, at, node);
        } else {
#pragma unroll
            for (int k = 0; k < WAYS; ++k) node[k] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(lnodes) + at[k]);
        }
        float val[WAYS];
#pragma unroll
        for (int k = 0; k < WAYS; ++k)                 // a leaf (var = 255) reads the last feature and ignores it
            val[k] = hist_at((int)(__umul24(min(node[k].y >> 24, last_var), xbytes) + xbase));
        unsigned long long all_at_leaf = every_lane;
#pragma unroll
        for (int k = 0; k < WAYS; ++k) {
            const bool leaf = (node[k].y >> 24) == 255u;
            all_at_leaf &= __ballot(leaf);
            if (STATS) depth += at[k] != rest ? 1 : 0;
            sum += leaf ? __uint_as_float(node[k].x) : 0.0f;             // (a small integer: exact in any order)
            uint32_t next;                                                // (child + right) x 8, the variable shifted out
            // of a leaf: on to the root of the next tree of the chain (its "left child")
            const bool left = leaf | (val[k] <= __uint_as_float(node[k].x));
            asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(next) : "v"(node[k].y), "v"(left ? 0u : 8u));
            at[k] = next & 0x07ffffffu;
        }
        return all_at_leaf;
    };
    for (;;) {
        // two steps per test (a resting walk stays at its leaf, so the second step's answer is enough): the test and the
        // compiler's loop bookkeeping around it are 10 instructions
        unsigned long long all_at_leaf;
        do {
            step();
            all_at_leaf = step();
        } while (all_at_leaf != every_lane);
        bool walking = false;
#pragma unroll
        for (int k = 0; k < WAYS; ++k) walking |= at[k] != rest;
        if (!__any(walking)) break;
    }
    return (int)sum;
}

// The scoring stage ("runForest", hpp:267-296) in two kernels over chunks of 64 consecutive storage
// positions of a view:
//   feature kernel -> feat[(chunk * F + c) * 64 + position in the chunk], one contiguous F x 64 block per
//                     chunk (kGroup waves of 64 / kGroup points write their columns of it)
//   forest kernel  <- the same block, one wave per chunk, staged in LDS next to the top of the forest.
// In one kernel the forest walk (dependent node reads, 64 different cache lines per load instruction)
// was 26 % of the time and ran no faster on its own at 5 waves per SIMD: it is bound by the texture
// path, not by occupancy (profiles/r02_notes.md).  On its own it can keep the forest in LDS.
struct WavePoint {
    int s;               // storage position
    bool in_range, scoreable;
    float4 p, np;
};

__device__ __forceinline__ WavePoint wave_point(const ViewDev &a, int chunk, int lane, bool want_xyz) {
    WavePoint w;
    w.s = chunk * kLanes + lane;
    const int nfinite = a.cell_start[a.ds->grid.ncells];
    w.in_range = w.s < nfinite;
    w.np = w.in_range ? a.nrm[w.s] : make_float4(0.f, 0.f, 0.f, 0.f);
    w.p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (w.in_range) {
        if (want_xyz) w.p = a.pts[w.s];
        else w.p.w = a.pts[w.s].w;                                                  // original index only
    }
    w.scoreable = w.in_range && w.np.w != 0.0f;                                    // hpp:277
    return w;
}

// Which view and which block of it a workgroup of a (blocks, views) grid takes.  Workgroups go to the 8 XCDs round robin by
// their linear number; `by_xcd` deals the VIEWS the same way -- with 8 views every XCD works on one view and its 4 MB of L2
// hold what that view reads at random (the sorted-search mode reads the caller's normals by original index: 2.4 MB per
// 200 k-point view, 19 MB for 8 views that every XCD touches otherwise).  Only for views of about the same size
// (launch_score): an XCD is not given work by how much it has left.
struct ViewBlock {
    unsigned view, bx;
};
__device__ __forceinline__ ViewBlock view_block(int by_xcd) {
    ViewBlock r;
    if (by_xcd) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
        r.view = lin % gridDim.y;
        r.bx = lin / gridDim.y;
    } else {
        r.view = blockIdx.y;
        r.bx = blockIdx.x;
    }
    return r;
}

// Sorted-search mode: which points take the path for LARGE neighborhoods (sorted_collect_kernel / sorted_add_kernel) without
// being searched here first.  A cheap test on the index alone -- the population of the point's own cell (one pair of
// cell_start[] entries): the box of a point holds about nine such cells on a surface, so more than kLargeCell points in the
// cell are about 170 neighbors or more, where the register sort of 128 keys is at its end.  The test only has to be
// roughly right: a point it lets through whose list runs full is deferred to the same kernels (point_features_sorted_view).  (Summing
// the candidates of the whole box -- 32 dependent loads per lane in a kernel with two waves per SIMD -- cost 7 % of the
// stage on views that have no large point at all.)
// A cell with more than kHugeCell points (~530 neighbors) is beyond what a wave sorts (kWaveKeys): such a point is listed for
// the workgroup kernel at once (top bit of its large_list entry + the second half of the list), sorted_collect_wave_kernel
// passes it over -- at the reference's own operating point (2 300 neighbors) every point is one of these.
constexpr int kLargeCell = 57, kHugeCell = 170;
constexpr unsigned kHugeBit = 0x80000000u;
__device__ __forceinline__ int own_cell_population(const GridDesc &g, const int *__restrict__ cell_start, const WavePoint &w) {
    if (!w.scoreable) return 0;
    const int cx = cell_coord(w.p.x, g.mn[0], g.h, g.dims[0]);
    const int cy = cell_coord(w.p.y, g.mn[1], g.h, g.dims[1]);
    const int cz = cell_coord(w.p.z, g.mn[2], g.h, g.dims[2]);
    const int c = (cz * g.dims[1] + cy) * g.dims[0] + cx;
    return cell_start[c + 1] - cell_start[c];
}

// Several independent views per launch (blockIdx.y = view): one 200 k-point view is only a few waves per
// SIMD; a batch of views fills the chip.  Workgroup (= wave) x handles the 64 / kGroup storage positions
// x * 64 / kGroup ..; it writes its columns of the F x 64 feature block of its chunk of 64 positions.
// K_f of the view for the handle's NEXT call (DevState::kf_sum / kf_points, cumulative; the host takes differences in
// kpl_sync_status and picks the walk and the lanes per point of the next launch from them -- api.cpp choose_walk): one wave
// in 64 adds the K_f of its points, a sample of every part of the view for two atomics per 64 waves
__device__ __forceinline__ bool kf_sampled(int bx) { return (bx & 63) == 0; }
// sum = what the lane adds (an exact count, or its share of a histogram's mass), points = 1 in the lanes that stand for a point
__device__ __forceinline__ void note_kf(const ViewDev &v, float sum, int points) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sum += __shfl_xor(sum, d);
        points += __shfl_xor(points, d);
    }
    if (threadIdx.x == 0 && points > 0) {
        atomicAdd(&v.ds->kf_sum, (unsigned long long)(sum + 0.5f));
        atomicAdd(&v.ds->kf_points, (unsigned long long)points);
    }
}

template <bool STATS, int G>
__global__ __launch_bounds__(kLanes) void feature_kernel(Batch b, int maxF, int ecap) {
    extern __shared__ float H[];
    constexpr int kPts = kLanes / G;
    const ViewDev &v = b.view[blockIdx.y];
    if (v.f.sorted || v.f.walk != kWalkLanes || v.f.lanes != G) return;       // scored by another kernel of the stage
    const int chunk = blockIdx.x / G, pi = threadIdx.x / G, gq = threadIdx.x % G;
    const int col = (blockIdx.x % G) * kPts + pi;                     // the point's column of the chunk's F x 64 block
    if (chunk * kLanes + col - pi >= v.n) return;
    const WavePoint w = wave_point(v, chunk, col, true);
    // every lane of the wave runs the feature code (wave-level votes inside); a group without a scoreable
    // point simply has no rows
    const bool sampled = kf_sampled(blockIdx.x);
    float mass = 0.0f;
    const int kf = point_features<G>(v.pts, v.nrm, v.cell_start, v.ds->grid, v.f, w.p, w.np, H,
                                     reinterpret_cast<uint2 *>(H + maxF * kPts), ecap, w.scoreable, sampled ? &mass : nullptr);
    if (STATS && w.scoreable && gq == 0) atomicAdd(&v.stats->sum_kf, (unsigned long long)kf);
    if (sampled) note_kf(v, w.scoreable ? mass + (gq == 0 ? 1.0f : 0.0f) : 0.0f, w.scoreable && gq == 0 ? 1 : 0);
    float *o = v.feat + (size_t)chunk * v.f.F * kLanes + col;
    for (int c = gq; c < v.f.F; c += G) o[c * kLanes] = H[c * kPts + pi];
}

// the same for the views whose neighborhoods are LARGE (FeatDesc::walk == kWalkTwoPass), first pass: the accept words of
// every point's whole walk -> its wave's block of the word list (search_point_words_staged)
constexpr int kSearchGroup = 8;        // lanes per point of the search pass: a word of 32 candidates is ONE step of the 8 lanes; 8 points per wave --
                                       // the waves of the densest cells walk 2.7 x the words of the average wave, the launch lasts as long as they do
constexpr int kSearchWindow = 128;     // candidates per staged window (two windows of 2 KB of LDS per wave: the ~31 waves per CU of a 63 k-point view are resident together)
// for_sorted: the launch of the sorted branch takes the views in sorted order (their lists feed sorted_words_kernel), the other
// one the views in canonical order (feature_drain_kernel)
template <bool STATS>
__global__ __launch_bounds__(kLanes) void feature_search_kernel(Batch b, int for_sorted) {
    __shared__ float4 sp[2 * kSearchWindow];        // two windows: one searched, the next on its way
    constexpr int G = kSearchGroup, kPts = kLanes / G;
    const ViewDev &v = b.view[blockIdx.y];
    if (v.f.walk != kWalkTwoPass || (v.f.sorted != 0) != (for_sorted != 0)) return;
    // (a view in sorted order keeps its key segments in sort_keys: its word lists have an array and tables of their own)
    uint2 *const wbase = v.f.sorted ? v.words : reinterpret_cast<uint2 *>(v.sort_keys);
    const unsigned long long wcap = v.f.sorted ? v.word_cap : v.key_cap;
    unsigned *const wstart = v.f.sorted ? v.wseg_start : v.seg_start;
    int *const wlen = v.f.sorted ? v.wseg_len : v.seg_len;
    const int chunk = blockIdx.x / G, pi = threadIdx.x / G, gq = threadIdx.x % G;
    const int col = (blockIdx.x % G) * kPts + pi;
    if (chunk * kLanes + col - pi >= v.n) return;
    const WavePoint w = wave_point(v, chunk, col, true);
    unsigned first = 0u;                           // the wave's block: first entry of its point 0
    bool fits = true;
    auto block = [&](int steps) -> uint2 * {
        // a 32nd of the array per cursor; the cursor of a wave = its number mod 32 (waves go to the XCDs round robin: four cursors per XCD)
        const unsigned long long share = wcap / kWordShards, need = (unsigned long long)steps * kPts;
        const int shard = blockIdx.x % kWordShards;
        unsigned long long off = 0ull;
        if (threadIdx.x == 0 && need > 0) off = atomicAdd(&v.ds->word_cursor[shard], need);
        off = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(off >> 32)) << 32) |
              (unsigned)__builtin_amdgcn_readfirstlane((int)off);
        fits = off + need <= share;
        if (!fits) {
            if (threadIdx.x == 0) atomicCAS(&v.ds->status, kStatusOk, kStatusKeyCapacity);
            return nullptr;
        }
        first = (unsigned)(share * shard + off);
        return wbase + first;
    };
    int entries = 0;
    const int kf = search_point_words_staged<G>(v.pts, v.cell_start, v.ds->grid, v.f, w.p, w.scoreable, sp, kSearchWindow, block, entries,
                                                !v.f.sorted);
    if (w.in_range && gq == 0) {
        wstart[w.s] = first + (unsigned)pi;
        wlen[w.s] = fits ? entries : 0;
    }
    if (STATS && !v.f.sorted && w.scoreable && gq == 0) atomicAdd(&v.stats->sum_kf, (unsigned long long)kf);     // (sorted: the kernels behind count)
    if (kf_sampled(blockIdx.x)) note_kf(v, w.scoreable && gq == 0 ? (float)kf : 0.0f, w.scoreable && gq == 0 ? 1 : 0);
}

// ... second pass: every point's list drained in one go (drain_point_words).  `stride` = points per wave of the search
// kernel (the lists of a wave's points are interleaved)
//   LDS: [H: maxF x 64 / G floats]
template <int G>
__global__ __launch_bounds__(kLanes) void feature_drain_kernel(Batch b, int maxF, int stride) {
    extern __shared__ float H[];
    constexpr int kPts = kLanes / G;
    const ViewDev &v = b.view[blockIdx.y];
    if (v.f.sorted || v.f.walk != kWalkTwoPass || v.f.lanes != G) return;
    const int chunk = blockIdx.x / G, pi = threadIdx.x / G, gq = threadIdx.x % G;
    const int col = (blockIdx.x % G) * kPts + pi;
    if (chunk * kLanes + col - pi >= v.n) return;
    const WavePoint w = wave_point(v, chunk, col, true);
    const int ecnt = w.scoreable ? v.seg_len[w.s] : 0;
    const uint2 *list = reinterpret_cast<const uint2 *>(v.sort_keys) + (w.scoreable ? v.seg_start[w.s] : 0u);
    drain_point_words<G>(v.pts, v.nrm, v.f, w.p, w.np, H, list, stride, ecnt);
    float *o = v.feat + (size_t)chunk * v.f.F * kLanes + col;
    for (int c = gq; c < v.f.F; c += G) o[c * kLanes] = H[c * kPts + pi];
}

// the same for the views in sorted-search mode: kSortGroup lanes per point, 64 / kSortGroup points per wave
//   LDS: [H: maxF x 16 floats][accept words: ecap x 16 uint2][position lists: lcap x 16 positions of 4 bytes]
template <bool STATS>
__global__ __launch_bounds__(kLanes) void feature_sorted_kernel(Batch b, int maxF, int ecap, int lcap, int by_xcd) {
    extern __shared__ float H[];
    constexpr int G = kSortGroup, kPts = kLanes / G;
    const ViewBlock vb = view_block(by_xcd);
    const ViewDev &v = b.view[vb.view];
    if (!v.f.sorted || v.f.walk == kWalkTwoPass) return;          // (walk 1: sorted_words_kernel)
    const int chunk = vb.bx / G, pi = threadIdx.x / G, gq = threadIdx.x % G;
    const int col = (vb.bx % G) * kPts + pi;
    if (chunk * kLanes + col - pi >= v.n) return;
    const WavePoint w = wave_point(v, chunk, col, true);
    // a point with a large neighborhood is scored by sorted_collect_kernel + sorted_add_kernel (below): no rows here, and
    // its column of the feature block is theirs
    const int own_cell = own_cell_population(v.ds->grid, v.cell_start, w);
    // (all_large: the handle's last call ended up listing nearly every point, most of them after a search that filled their
    // register lists for nothing -- 1.5 of 7.6 ms for 8 x 200 k points at 10 mesh resolutions)
    // (all_large == 2: the handle's last call stored more than a thousand keys per listed point -- the reference's default radius:
    // 2 300 --: every point is for the workgroup kernel, the wave-per-point kernel is not even launched)
    const bool large = own_cell > kLargeCell || (v.f.all_large && w.scoreable), huge = own_cell > kHugeCell || (v.f.all_large == 2 && w.scoreable);
    uint2 *ent = reinterpret_cast<uint2 *>(H + maxF * kPts);
    unsigned *tl = reinterpret_cast<unsigned *>(ent + ecap * kPts);
    bool deferred = false, overflow = false;
    int kf = 0;
    if (__any(w.scoreable && !large))           // (a wave whose points are all large only lists them)
        kf = point_features_sorted_view<G>(v.pts, v.nrm, v.cell_start, v.ds->grid, v.f, w.p, w.np, H, ent, ecap,
                                           tl, lcap, w.scoreable && !large, deferred, overflow);
    else
        for (int c = gq; c < v.f.F; c += G) H[c * kPts + pi] = 0.0f;      // (what a point that is not scored leaves in its column)
    // the longest neighborhood of the wave (a deferred point: "longer than the list") for the list capacity of the handle's
    // next launch: a plain read of the running maximum first, the atomic only when the wave raises it (a handful per launch)
    {
        int m = (w.scoreable && !large) ? (overflow ? kSortedListKeys + 1 : kf) : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
        if (threadIdx.x == 0 && m > __hip_atomic_load(&v.ds->kf_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&v.ds->kf_max, m);
    }
    // a large box, or more neighbors than the list holds: a point for the collect / add kernels.  ONE atomic per wave for its
    // (up to 16) listed points: returning atomics on one address complete one after the other, ~18 ns each -- a view whose
    // 200 k points all went to the list one by one spent 3.7 ms on nothing else (profiles/r04_notes.md)
    const bool listed = deferred || large;
    const unsigned long long lbal = __ballot(listed && gq == 0);
    if (lbal != 0ull) {
        const int lane = threadIdx.x;
        int base = 0;
        if (lane == __builtin_ctzll(lbal)) base = atomicAdd(&v.ds->large_count, __popcll(lbal));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(lbal));
        if (listed && gq == 0) v.large_list[base + __popcll(lbal & ((1ull << lane) - 1ull))] = (int)((unsigned)w.s | (huge ? kHugeBit : 0u));
        const unsigned long long hbal = __ballot(huge && gq == 0);
        if (hbal != 0ull) {
            int hbase = 0;
            if (lane == __builtin_ctzll(hbal)) hbase = atomicAdd(&v.ds->huge_count, __popcll(hbal));
            hbase = __builtin_amdgcn_readlane(hbase, __builtin_ctzll(hbal));
            if (huge && gq == 0) v.large_list[v.n + hbase + __popcll(hbal & ((1ull << lane) - 1ull))] = w.s;
        }
    }
    if (listed) return;
    if (STATS && w.scoreable && gq == 0) atomicAdd(&v.stats->sum_kf, (unsigned long long)kf);
    float *o = v.feat + (size_t)chunk * v.f.F * kLanes + col;
    for (int c = gq; c < v.f.F; c += G) o[c * kLanes] = H[c * kPts + pi];
}

// sorted order through the word lists (point_features_sorted_words): kWordsGroup lanes per point, 8 points per wave
//   LDS: [H: maxF x 8 floats][position lists: kWordsKeys x 8 positions]
template <bool STATS, int EMAX>
__global__ __launch_bounds__(kLanes) void sorted_words_kernel(Batch b, int maxF, int by_xcd) {
    extern __shared__ float H[];
    constexpr int G = kWordsGroup, kPts = kLanes / G;
    const ViewBlock vb = view_block(by_xcd);
    const ViewDev &v = b.view[vb.view];
    if (!v.f.sorted || v.f.walk != kWalkTwoPass || (v.f.lcap > 32 * G ? 64 : 32) != EMAX) return;
    const int chunk = vb.bx / G, pi = threadIdx.x / G, gq = threadIdx.x % G;
    const int col = (vb.bx % G) * kPts + pi;
    if (chunk * kLanes + col - pi >= v.n) return;
    const WavePoint w = wave_point(v, chunk, col, true);
    unsigned *tl = reinterpret_cast<unsigned *>(H + maxF * kPts);
    const int ecnt = w.scoreable ? v.wseg_len[w.s] : 0;
    const uint2 *list = v.words + (w.scoreable ? v.wseg_start[w.s] : 0u);
    bool deferred = false;
    const int kf = point_features_sorted_words<G, EMAX>(v.pts, v.nrm, v.f, w.p, w.np, H, tl, list, kLanes / kSearchGroup, ecnt, deferred);
    // a point whose list ran full: for the wave-per-point kernel (and the workgroup kernel behind it), as in feature_sorted_kernel
    const unsigned long long lbal = __ballot(deferred && gq == 0);
    if (lbal != 0ull) {
        const int lane = threadIdx.x;
        int base = 0;
        if (lane == __builtin_ctzll(lbal)) base = atomicAdd(&v.ds->large_count, __popcll(lbal));
        base = __builtin_amdgcn_readlane(base, __builtin_ctzll(lbal));
        if (deferred && gq == 0) v.large_list[base + __popcll(lbal & ((1ull << lane) - 1ull))] = w.s;
    }
    if (deferred) return;              // (the mean neighborhood of the view is sampled by the search pass, deferred points included)
    if (STATS && w.scoreable && gq == 0) atomicAdd(&v.stats->sum_kf, (unsigned long long)kf);
    float *o = v.feat + (size_t)chunk * v.f.F * kLanes + col;
    for (int c = gq; c < v.f.F; c += G) o[c * kLanes] = H[c * kPts + pi];
}

// Persistent workgroups of several waves: the workgroup stages the first nlds nodes of its view's
// forest once, then every wave takes chunks of 64 points: features -> its LDS slice, tree walks,
// score_sorted / scores / NMS candidates out.
//   LDS: [nlds_cap nodes, 8 B each][waves x F x 64 floats]
template <bool STATS>
__global__ __launch_bounds__(1024) void forest_kernel(Batch b, int maxF, int nlds_cap) {
    extern __shared__ uint2 lnodes[];
    const ViewDev &a = b.view[blockIdx.y];
    const int lane = threadIdx.x & (kLanes - 1), wid = threadIdx.x / kLanes, nwaves = blockDim.x / kLanes;
    const int nlds = min(nlds_cap, a.forest.ntop);        // the top part, or as much of it as fits
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) lnodes[i] = a.forest.nodes[i];
    __syncthreads();
    float *H = reinterpret_cast<float *>(lnodes + nlds_cap) + (size_t)wid * maxF * kLanes;
    const int nchunks = (a.n + kLanes - 1) / kLanes;
    const int F = a.f.F, stride = gridDim.x * nwaves;
    // the feature block of the NEXT chunk of this wave is requested before the trees of the current one are
    // walked (up to kFeatAhead floats per lane in registers; larger histograms load the rest at the chunk's
    // turn): a wave would otherwise sit out a trip to memory at the start of every chunk
    constexpr int kFeatAhead = 32;
    float ahead[kFeatAhead];
    auto request = [&](int chunk) {
        const float *o = a.feat + (size_t)chunk * F * kLanes + lane;
#pragma unroll
        for (int c = 0; c < kFeatAhead; ++c) ahead[c] = (chunk < nchunks && c < F) ? o[c * kLanes] : 0.0f;
    };
    int chunk = blockIdx.x * nwaves + wid;
    request(chunk);
    for (; chunk < nchunks; chunk += stride) {
        {   // per ORIGINAL point: NaN for points that are not in the grid
            const int i = chunk * kLanes + lane;
            if (i < a.n && a.scores && a.cid[i] < 0) a.scores[i] = NAN;
        }
        const WavePoint w = wave_point(a, chunk, lane, false);
#pragma unroll
        for (int c = 0; c < kFeatAhead; ++c)
            if (c < F) H[c * kLanes + lane] = ahead[c];
        if (F > kFeatAhead) {
            const float *o = a.feat + (size_t)chunk * F * kLanes + lane;
            for (int c = kFeatAhead; c < F; ++c) H[c * kLanes + lane] = o[c * kLanes];
        }
        request(chunk + stride);
        if (!w.in_range) continue;
        float score = NAN;
        if (w.scoreable) {
            int depth = 0;
            float fsum;
            if (nlds < a.forest.nnodes)                      // blocks below the top part (or a top part beyond the LDS budget)
                fsum = forest_sum_deep<STATS>(a.forest, lnodes, nlds, H + lane, depth);
            // trees out of step only where it pays (more trees than ways) and is exact (integer leaves)
            else if (a.forest.order_free && a.forest.ntrees > kTreeWays)
                fsum = (float)forest_sum_any_order<STATS, kTreeWays>(a.forest, lnodes, nlds, H + lane, kLanes, F, 0, 1, true, depth);
            else
                fsum = forest_sum<STATS>(a.forest, lnodes, nlds, H + lane, F, depth);
            score = 1 - (fsum / (a.forest.ntrees * 1.0f));                         // hpp:287
            if (STATS) {
                atomicAdd(&a.stats->sum_depth, (unsigned long long)depth);
                atomicAdd(&a.stats->n_scored, 1ull);
            }
        }
        a.score_sorted[w.s] = score;
        if (a.scores) a.scores[__float_as_int(w.p.w)] = score;
        // hand the point to the NMS stage (detectKeypoints, hpp:203-208): only scoreable points whose
        // score, promoted to double, is not below the threshold are ever searched
        if (w.scoreable) {
            if (!a.nd.non_maxima) a.flags[__float_as_int(w.p.w)] = 1;               // hpp:189-196
            // hpp:205-207: a non-finite response is never a candidate (!pcl_isfinite(intensity))
            else if (isfinite(score) && !((double)score < a.nd.thr)) a.cand.list[atomicAdd(a.cand.count, 1)] = w.s;
        }
    }
}

// The forest kernel for small forests whose sum is exact in any order (class labels: every forest the reference
// trains; the 10-tree bench forest): G = 2 lanes per point.  A wave takes half a chunk (32 points, F x 32 floats = 3.8 KB
// at F = 30 instead of 7.7 KB), lane 32 g + p walks the trees g, g + 2, ... of point p in step (5 or 8 of them), the
// partial sums meet through one cross-lane read.  Same work per point; but a wave's in-step walk lasts as long as the
// deepest of its walks (320 instead of 640 of them), and 16 waves need 118 KB of LDS instead of 13 waves 156 KB: the other
// batch's feature kernel keeps 40 KB of every CU while this one runs.  8 views of 200 k points: 0.152 -> 0.133 ms alone,
// two batches in flight 1 802 -> 1 884 Mpoints/s.
constexpr int kPairWays = 8;           // trees per lane at most: forests of up to 16 trees (5 for up to 10)
template <bool STATS, int G, int WAYS>
__global__ __launch_bounds__(1024) void forest_pair_kernel(Batch b, int maxF, int nlds_cap) {
    extern __shared__ uint2 lnodes[];
    constexpr int kPts = kLanes / G;                         // points per wave
    const ViewDev &a = b.view[blockIdx.y];
    const int lane = threadIdx.x & (kLanes - 1), wid = threadIdx.x / kLanes, nwaves = blockDim.x / kLanes;
    const int p = lane & (kPts - 1), g = lane / kPts;
    const int nlds = min(nlds_cap, a.forest.nnodes);        // the whole forest (launch_forest_stage checks that it fits)
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) lnodes[i] = a.forest.nodes[i];
    __syncthreads();
    float *H = reinterpret_cast<float *>(lnodes + nlds_cap) + (size_t)wid * maxF * kPts;
    const int nunits = (a.n + kPts - 1) / kPts;              // units of kPts consecutive storage positions
    const int F = a.f.F, stride = gridDim.x * nwaves;
    const int nfinite = a.cell_start[a.ds->grid.ncells];
    // the feature rows of the NEXT unit are requested before the trees of the current one are walked: rows G c + g of
    // the unit per load (G rows per load instruction)
    constexpr int kAhead = 32 / G;
    float ahead[kAhead];
    auto request = [&](int unit) {
        const float *o = a.feat + (size_t)(unit / G) * F * kLanes + (unit % G) * kPts + p;
#pragma unroll
        for (int c = 0; c < kAhead; ++c) ahead[c] = (unit < nunits && G * c + g < F) ? o[(G * c + g) * kLanes] : 0.0f;
    };
    int unit = blockIdx.x * nwaves + wid;
    request(unit);
    for (; unit < nunits; unit += stride) {
        {   // per ORIGINAL point: NaN for points that are not in the grid
            const int i = unit * kPts + p;
            if (g == 0 && i < a.n && a.scores && a.cid[i] < 0) a.scores[i] = NAN;
        }
        const int s = unit * kPts + p;                       // storage position of the lane's point
        const bool in_range = s < nfinite;
        const float4 np = in_range ? a.nrm[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const bool scoreable = in_range && np.w != 0.0f;     // hpp:277
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < kAhead; ++c)
            if (G * c + g < F) H[(G * c + g) * kPts + p] = ahead[c];
        if (F > G * kAhead) {
            const float *o = a.feat + (size_t)(unit / G) * F * kLanes + (unit % G) * kPts + p;
            for (int c = G * kAhead + g; c < F; c += G) H[c * kPts + p] = o[c * kLanes];
        }
        wave_lds_fence();
        request(unit + stride);
        int depth = 0;
        float part = forest_sum_strided<STATS, WAYS>(a.forest, lnodes, H + p, kPts, F, g, G, depth);
#pragma unroll
        for (int off = kPts; off < kLanes; off <<= 1) {      // exact in any order: integer leaf values
            part += __shfl_xor(part, off);
            if (STATS) depth += __shfl_xor(depth, off);
        }
        if (g != 0 || !in_range) continue;
        float score = NAN;
        if (scoreable) {
            score = 1 - (part / (a.forest.ntrees * 1.0f));                          // hpp:287
            if (STATS) {
                atomicAdd(&a.stats->sum_depth, (unsigned long long)depth);
                atomicAdd(&a.stats->n_scored, 1ull);
            }
        }
        const int orig = __float_as_int(a.pts[s].w);
        a.score_sorted[s] = score;
        if (a.scores) a.scores[orig] = score;
        if (scoreable) {
            if (!a.nd.non_maxima) a.flags[orig] = 1;                                // hpp:189-196
            else if (isfinite(score) && !((double)score < a.nd.thr)) a.cand.list[atomicAdd(a.cand.count, 1)] = s;
        }
    }
}

// The forest kernel for a large histogram and many trees whose sum is exact in any order (config 5:
// F = 80, 100 trees, 16 MB of nodes; the forest in the chained layout of forest.h): 64 / G points per wave, G lanes
// per point, lane g of a point walks the chains g, g + G, ... (forest_sum_chained) and the G partial sums are added at
// the end.  The F x 64 floats of a whole wave of points (20 KB at F = 80) leave room for 4 waves per CU next to the node
// cache -- one per SIMD, and every step of the walk waits for a node from beyond the L2; with G = 4 the slice of a
// wave is 5 KB and 16 waves fit.

template <bool STATS>
__global__ __launch_bounds__(1024) void forest_split_kernel(Batch b, int maxF, int nlds_cap, int G) {
    extern __shared__ uint2 lnodes[];
    const ViewDev &a = b.view[blockIdx.y];
    const int lane = threadIdx.x & (kLanes - 1), wid = threadIdx.x / kLanes, nwaves = blockDim.x / kLanes;
    const int ppw = kLanes / G, p = lane % ppw, g = lane / ppw;
    const int nlds = min(nlds_cap, a.forest.ntop);        // the top part, or as much of it as fits
    if (lds_address(reinterpret_cast<const float *>(lnodes)) != 0) __builtin_trap();      // fetch_nodes_masked relies on it
    for (int i = threadIdx.x; i < nlds; i += blockDim.x) lnodes[i] = a.forest.nodes[i];
    __syncthreads();
    float *H = reinterpret_cast<float *>(lnodes + nlds_cap) + (size_t)wid * maxF * ppw;
    const int F = a.f.F;
    const int nunits = (a.n + ppw - 1) / ppw;
    const int nfinite = a.cell_start[a.ds->grid.ncells];
    for (int unit = blockIdx.x * nwaves + wid; unit < nunits; unit += gridDim.x * nwaves) {
        const int s = unit * ppw + p;                              // storage position of the lane's point
        if (g == 0 && s < a.n && a.scores && a.cid[s] < 0) a.scores[s] = NAN;   // (s as an ORIGINAL index here)
        const bool in_range = s < nfinite;
        const float4 np = in_range ? a.nrm[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const bool scoreable = in_range && np.w != 0.0f;           // hpp:277
        // the wave's ppw feature rows: element (c, point) of the F x 64 block the feature kernel wrote
        wave_lds_fence();
        for (int idx = lane; idx < F * ppw; idx += kLanes) {
            const int c = idx / ppw, sp = unit * ppw + idx % ppw;
            H[idx] = sp < nfinite ? a.feat[((size_t)(sp / kLanes) * F + c) * kLanes + sp % kLanes] : 0.0f;
        }
        wave_lds_fence();
        int depth = 0;
        int sum = nlds < a.forest.nnodes
                      ? forest_sum_chained<STATS, true>(a.forest, lnodes, nlds, H + p, ppw, F, g, G, scoreable, depth)
                      : forest_sum_chained<STATS, false>(a.forest, lnodes, nlds, H + p, ppw, F, g, G, scoreable, depth);
        for (int off = ppw; off < kLanes; off <<= 1) {
            sum += __shfl_xor(sum, off);
            if (STATS) depth += __shfl_xor(depth, off);
        }
        if (g != 0 || !in_range) continue;
        float score = NAN;
        if (scoreable) {
            score = 1 - ((float)sum / (a.forest.ntrees * 1.0f));                   // hpp:287
            if (STATS) {
                atomicAdd(&a.stats->sum_depth, (unsigned long long)depth);
                atomicAdd(&a.stats->n_scored, 1ull);
            }
        }
        const int orig = __float_as_int(a.pts[s].w);
        a.score_sorted[s] = score;
        if (a.scores) a.scores[orig] = score;
        if (scoreable) {
            if (!a.nd.non_maxima) a.flags[orig] = 1;                                // hpp:189-196
            else if (isfinite(score) && !((double)score < a.nd.thr)) a.cand.list[atomicAdd(a.cand.count, 1)] = s;
        }
    }
}

// computePointsForTrainingFeatures, hpp:299-318: same feature code, sparse query lists (64 / kGroup queries per wave) of up to
// 8 views per launch (blockIdx.y = view: the training set of /root/reference/src/main_train_detector.cpp:413-446 is a few
// hundred points in each of many views -- one view alone is a handful of waves)
__global__ __launch_bounds__(kLanes) void features_kernel(QueryBatch qb, int maxF, int ecap) {
    extern __shared__ float H[];
    constexpr int kPts = kLanes / kGroup;
    const QueryView &q = qb.view[blockIdx.y];
    if (q.f.sorted) return;                        // features_sorted_kernel
    const int pi = threadIdx.x / kGroup, gq = threadIdx.x % kGroup;
    const int qi = blockIdx.x * kPts + pi;
    if (qi - pi >= q.m) return;
    uint2 *ent = reinterpret_cast<uint2 *>(H + maxF * kPts);
    const GridDesc g = q.ds->grid;
    int s = -1;
    if (qi < q.m) {
        const int i = q.query[qi];
        s = (i >= 0 && i < q.n) ? q.pos_of[i] : -1;
    }
    const float4 p = s >= 0 ? q.pts[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 np = s >= 0 ? q.nrm[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    point_features<kGroup>(q.pts, q.nrm, q.cell_start, g, q.f, p, np, H, ent, ecap, s >= 0);
    if (qi >= q.m) return;
    float *o = q.out + (size_t)qi * q.f.F;
    for (int c = gq; c < q.f.F; c += kGroup) o[c] = s >= 0 ? H[c * kPts + pi] : NAN;
}

// the same in sorted-search mode (nrmsrc: the caller's normals in original order, byte stride ns)
__global__ __launch_bounds__(kLanes) void features_sorted_kernel(QueryBatch qb, int maxF, int ecap, int lcap) {
    extern __shared__ float H[];
    constexpr int G = kSortGroup, kPts = kLanes / G;
    const QueryView &q = qb.view[blockIdx.y];
    if (!q.f.sorted) return;
    const int pi = threadIdx.x / G, gq = threadIdx.x % G;
    const int qi = blockIdx.x * kPts + pi;
    if (qi - pi >= q.m) return;
    uint2 *ent = reinterpret_cast<uint2 *>(H + maxF * kPts);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(ent + ecap * kPts);
    const GridDesc g = q.ds->grid;
    int s = -1;
    if (qi < q.m) {
        const int i = q.query[qi];
        s = (i >= 0 && i < q.n) ? q.pos_of[i] : -1;
    }
    const float4 p = s >= 0 ? q.pts[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 np = s >= 0 ? q.nrm[s] : make_float4(0.f, 0.f, 0.f, 0.f);
    point_features_sorted<G>(q.pts, q.nrmsrc, q.ns, q.cell_start, g, q.f, p, np, H, ent, ecap, keys, lcap, s >= 0);
    if (qi >= q.m) return;
    float *o = q.out + (size_t)qi * q.f.F;
    for (int c = gq; c < q.f.F; c += G) o[c] = s >= 0 ? H[c * kPts + pi] : NAN;
}

// ---------------------------------------------------------------------------------------------
// Sorted-search mode, LARGE neighborhoods (the reference's own default operating point: radiusFeatures 20 on the cheff
// views = 30 mesh resolutions, K_f ~ 2 300 -- /root/reference/src/main_test_detector.cpp:65).  The register sort of
// point_features_sorted holds 128 keys per point and pass and searches again for every pass: 20 passes over ~10 000 candidates
// per point there (46 ms per 63 k-point view against 1.8 ms in the canonical order).  Points in crowded cells (own_cell_population > kLargeCell)
// and points whose register list ran full take three kernels instead:
//   feature_sorted_kernel  lists them instead of scoring them (DevState::large_count, ViewDev::large_list), and with them
//                          the points whose list ran full, or whose order the stand-ins do not decide
//   sorted_collect_kernel  ONE WORKGROUP PER POINT: its 256 threads walk the rows of the box together -- consecutive
//                          storage positions, coalesced 16-byte loads, no lock step with other points --, the accepted
//                          neighbors' keys (d2 bits << 32 | original index) go to a list in LDS, the list is sorted there
//                          (one counting pass over 1024 buckets linear in d2 -- a surface has about equally many neighbors
//                          per unit of d2 --, then every thread orders its four buckets by insertion; lists that defeat the
//                          buckets -- many equal distances -- take a bitonic network), and the sorted keys are written to the
//                          point's segment of ViewDev::sort_keys (bump allocation: one atomic per point).  A list longer
//                          than the LDS holds (4096 keys) is cut into windows of d2, each collected, sorted and appended.
//   sorted_add_kernel      the feature loop proper (hpp:334-359) over the sorted segment: two lanes per point, the
//                          contributions of two neighbors per round computed side by side, the histogram updates in order.
// Same keys, same arithmetic, same order as point_features_sorted: the results are the same bits.
// Segments that do not fit ViewDev::key_cap set kStatusKeyCapacity: the call fails with KPL_ERR_RETRY after
// kpl_sync_status has grown the array to DevState::keys_needed.
// ---------------------------------------------------------------------------------------------
constexpr int kCollectThreads = 256, kCollectKeys = 4096, kCollectBuckets = 1024, kCollectAhead = 4;
constexpr int kInsertionMax = 48;        // keys a thread orders by insertion; more in its buckets: the whole list by the network

// ---- neighborhoods of up to kWaveKeys keys: ONE WAVE PER POINT (sorted_collect_wave_kernel) -------------------------------
// Between what the register lists of feature_sorted_kernel hold (124 keys) and the neighborhoods the workgroup kernel below
// is built for (thousands) a workgroup per point is mostly fixed cost: barriers per piece, a 1 024-bucket counting pass
// for 200 keys (8 views of 200 k points at 10 mesh resolutions, K_f = 190: 14 ms of collect against 1.4 ms for the whole
// canonical feature kernel).  Here every wave takes points of its own: the rows of the box 128 candidates at a time (the next
// piece in flight), the accepted keys appended to the wave's list in LDS by ballot, the list then sorted IN REGISTERS -- E =
// 1, 2, 4 or 8 keys per lane (blocked: lane l holds elements E l .. E l + E - 1), a bitonic network whose partners inside a
// lane are registers, inside a row of 16 lanes come through DPP moves and across rows through ds_bpermute -- and written to
// the point's segment of sort_keys.  A point
// with more candidates in its box than kWaveCandidates, or more than kWaveKeys accepted, goes to the second half of
// large_list (DevState::huge_count) for the workgroup kernel.
constexpr int kWaveKeys = 512, kWaveCollectWaves = 4, kWaveAhead = 2;
constexpr unsigned kWaveChunk = 2048;               // keys a wave takes from the key array at a time (16 KB)
constexpr int kWaveCandidates = 6 * kWaveKeys;       // ~4.4 candidates per neighbor on a surface: beyond, the list would not hold them

// the value of lane l ^ D: DPP moves inside a row of 16 lanes (no LDS crossbar, no wait), ds_bpermute across rows
template <int D>
__device__ __forceinline__ unsigned lane_xor32(unsigned x) {
    if (D == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
    if (D == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
    if (D == 4) {       // banks 0 and 2 of a row (lanes 0-3, 8-11) read 4 lanes up (row_shl:4), banks 1 and 3 read 4 lanes down (row_shr:4)
        const int t = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xf, 0xa, false);
    }
    if (D == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);      // row_ror:8
    return (unsigned)__shfl_xor((int)x, D);
}
template <int D>
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long x) {
    return ((unsigned long long)lane_xor32<D>((unsigned)(x >> 32)) << 32) | lane_xor32<D>((unsigned)x);
}

// (d is a constant wherever the network's loops are unrolled: the switch folds)
__device__ __forceinline__ unsigned long long lane_xor64(unsigned long long x, int d) {
    switch (d) {
    case 1: return lane_xor64<1>(x);
    case 2: return lane_xor64<2>(x);
    case 4: return lane_xor64<4>(x);
    case 8: return lane_xor64<8>(x);
    case 16: return lane_xor64<16>(x);
    default: return lane_xor64<32>(x);
    }
}

// (as sort_key_lists: the network runs on 32-bit stand-ins  q(d2) << SB | slot  of the keys -- min / max / median-of-three
// instead of 64-bit compares and selects, one lane exchange per comparator instead of two -- and the keys are fetched through
// the slot numbers; a list in which two neighbours of the sorted stand-ins share a q is sorted by its 64-bit keys)
template <int E>
__device__ __forceinline__ void wave_sort_store(const unsigned long long *list, int n, int lane, float r2,
                                                unsigned long long *__restrict__ out) {
    constexpr int SB = E == 1 ? 6 : E == 2 ? 7 : E == 4 ? 8 : 9;
    static_assert((1 << SB) == kWave * E, "slot bits");
    constexpr unsigned kQEnd = 1u << (31 - SB);
    const float scale = pin_f((float)kQEnd * 0.999f / r2);
    if (scale < 3.0e38f) {
        unsigned r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned i = (unsigned)(lane * E + e);
            const float d2 = __uint_as_float(reinterpret_cast<const unsigned *>(list)[2 * max(min((int)i, n - 1), 0) + 1]);
            const unsigned q = (unsigned)(d2 * scale);
            r[e] = (((int)i < n ? q : kQEnd + 1u + i) << SB) | i;
        }
#pragma unroll
        for (int k = 2; k <= kWave * E; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j < E) {
                    const bool lane_up = (lane & (k / E)) == 0;        // (k >= E)
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & j) != 0) continue;
                        const unsigned a = r[e], c = r[e | j];
                        const unsigned lo = min(a, c), hi = max(a, c);
                        const bool up = k < E ? (e & k) == 0 : lane_up;
                        r[e] = up ? lo : hi;
                        r[e | j] = up ? hi : lo;
                    }
                } else {
                    const bool keep_min = ((lane & (j / E)) == 0) == ((lane & (k / E)) == 0);
                    const unsigned bound = keep_min ? 0u : ~0u;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        unsigned o;
                        switch (j / E) {
                        case 1: o = lane_xor32<1>(r[e]); break;
                        case 2: o = lane_xor32<2>(r[e]); break;
                        case 4: o = lane_xor32<4>(r[e]); break;
                        case 8: o = lane_xor32<8>(r[e]); break;
                        case 16: o = lane_xor32<16>(r[e]); break;
                        default: o = lane_xor32<32>(r[e]); break;
                        }
                        unsigned res;
                        asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(res) : "v"(r[e]), "v"(o), "v"(bound));
                        r[e] = res;
                    }
                }
            }
        }
        unsigned near = ~0u;
#pragma unroll
        for (int e = 0; e + 1 < E; ++e) near = min(near, r[e] ^ r[e + 1]);
        const unsigned nxt = (unsigned)__shfl_down((int)r[0], 1);
        if (lane != kWave - 1) near = min(near, r[E - 1] ^ nxt);
        if (!__any((near >> SB) == 0u)) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = lane * E + e;
                if (i < n) out[i] = list[r[e] & (unsigned)(kWave * E - 1)];
            }
            return;
        }
    }
    unsigned long long r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        r[e] = i < n ? list[i] : ~0ull;
    }
#pragma unroll
    for (int k = 2; k <= kWave * E; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {                        // partners inside the lane: elements e and e | j
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if ((e & j) != 0) continue;
                    const bool up = k < E ? (e & k) == 0 : (lane & (k / E)) == 0;
                    const unsigned long long a = r[e], c = r[e | j];
                    const bool sw = (a > c) == up;
                    r[e] = sw ? c : a;
                    r[e | j] = sw ? a : c;
                }
            } else {                            // partner lane l ^ (j / E), the same element; the lower lane keeps the minimum when ascending
                const bool keep_min = ((lane & (j / E)) == 0) == ((lane & (k / E)) == 0);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const unsigned long long o = lane_xor64(r[e], j / E);
                    r[e] = ((o < r[e]) == keep_min) ? o : r[e];
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = lane * E + e;
        if (i < n) out[i] = r[e];
    }
}

__global__ __launch_bounds__(kWaveCollectWaves *kWave) void sorted_collect_wave_kernel(Batch b, int by_xcd) {
    __shared__ unsigned long long lists[kWaveCollectWaves][kWaveKeys];
    const ViewBlock vb = view_block(by_xcd);
    const ViewDev &v = b.view[vb.view];
    if (!v.f.sorted) return;
    DevState *ds = v.ds;
    const int nlarge = ds->large_count;
    // every listed point is one for the workgroup kernel already (the count only grows in this kernel, by points that were not:
    // equality means there are none)
    if (__hip_atomic_load(&ds->huge_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nlarge) return;
    const GridDesc g = ds->grid;
    const float4 *__restrict__ pts = v.pts;
    const int *__restrict__ cell_start = v.cell_start;
    int *__restrict__ huge_list = v.large_list + v.n;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    unsigned long long *list = lists[wid];
    const float r2 = v.f.r2;
    const int nwaves = gridDim.x * kWaveCollectWaves;
    // first position and end of row r of the box of p in lane r < 16 (empty where the box has none)
    auto box_rows = [&](const float4 &p, bool valid_point, int &row0, int &row1) {
        row0 = row1 = 0;
        CellBox bx = make_box(g, p.x, p.y, p.z, v.f.rr);
        bx.hi[1] = min(bx.hi[1], bx.lo[1] + 3);
        bx.hi[2] = min(bx.hi[2], bx.lo[2] + 3);
        if (valid_point && lane < 16) {
            const int z = bx.lo[2] + lane / 4, y = bx.lo[1] + lane % 4;
            if (z <= bx.hi[2] && y <= bx.hi[1]) {
                const int row = (z * g.dims[1] + y) * g.dims[0];
                row0 = cell_start[row + bx.lo[0]];
                row1 = cell_start[row + bx.hi[0] + 1];
            }
        }
    };
    // The chain list entry -> point -> row bounds of the NEXT point is requested while the current one is collected and sorted
    // (three dependent round trips per point otherwise, with five waves per SIMD to hide them).
    // Key segments come out of chunks of kWaveChunk keys that the wave takes from DevState::key_cursor with one atomic each
    // (one atomic per POINT on that one address was 3.6 of the kernel's 5 ms at 200 keys per point); what is left of a chunk
    // when the next list does not fit is not used (counted in key_cursor, like everything handed out).  With few points per
    // wave a point takes exactly its keys: nothing to gain there, and small arrays are not wasted on.
    const unsigned chunk_keys = nlarge >= 8 * nwaves ? kWaveChunk : 0u;
    unsigned long long chunk_pos = 0ull, chunk_end = 0ull;
    int li = vb.bx * kWaveCollectWaves + wid;
    // (list entries: the storage position, top bit set = listed for the workgroup kernel already, passed over here; -1 past
    // the end of the list)
    constexpr int kEnd = -1;
    auto is_mine = [](int e) { return e != kEnd && ((unsigned)e & kHugeBit) == 0u; };
    auto position = [](int e) { return (int)((unsigned)e & ~kHugeBit); };
    int s_cur = li < nlarge ? v.large_list[li] : kEnd;
    int s_nxt = li + nwaves < nlarge ? v.large_list[li + nwaves] : kEnd;
    float4 p = pts[s_cur == kEnd ? 0 : position(s_cur)];
    int my_r0, my_r1;
    box_rows(p, is_mine(s_cur), my_r0, my_r1);
    while (s_cur != kEnd) {
        const int s = s_cur;
        const int li_nn = li + 2 * nwaves;
        const int s_nn = li_nn < nlarge ? v.large_list[li_nn] : kEnd;
        const float4 p_nxt = pts[s_nxt == kEnd ? 0 : position(s_nxt)];
        int cands = my_r1 - my_r0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) cands += __shfl_xor(cands, off);
        cands = __builtin_amdgcn_readfirstlane(cands);
        int cnt = 0;
        const bool mine = is_mine(s);           // (uniform)
        if (mine && cands <= kWaveCandidates) {
            wave_lds_fence();                   // (the list of the point before has been read into registers)
            constexpr int kPiece = kWave * kWaveAhead;
            int row = -1, t0 = 0, r1 = 0;       // (uniform) the piece [t0, min(t0 + kPiece, r1)) of row `row`
            auto advance = [&]() -> bool {
                t0 += kPiece;
                while (t0 >= r1) {
                    if (++row >= 16) return false;
                    t0 = __builtin_amdgcn_readlane(my_r0, row);
                    r1 = __builtin_amdgcn_readlane(my_r1, row);
                }
                return true;
            };
            auto load_piece = [&](float4 (&q)[kWaveAhead]) {
#pragma unroll
                for (int a = 0; a < kWaveAhead; ++a) q[a] = pts[min(t0 + a * kWave + lane, r1 - 1)];
            };
            float4 cur[kWaveAhead], nxt[kWaveAhead];
            bool have = advance();
            if (have) load_piece(cur);
            while (have) {
                const int c_t0 = t0, c_r1 = r1;
                const bool have_next = advance();
                if (have_next) load_piece(nxt);
#pragma unroll
                for (int a = 0; a < kWaveAhead; ++a) {
                    const float d2 = dist2(p.x, p.y, p.z, cur[a]);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(unsigned)__float_as_int(cur[a].w);
                    const bool take = c_t0 + a * kWave + lane < c_r1 && d2 < r2;                    // strict (KdTreeFLANN)
                    const unsigned long long bal = __ballot(take);
                    const int slot = cnt + __popcll(bal & ((1ull << lane) - 1ull));
                    if (take && slot < kWaveKeys) list[slot] = key;
                    cnt += __popcll(bal);
                }
#pragma unroll
                for (int a = 0; a < kWaveAhead; ++a) cur[a] = nxt[a];
                have = have_next && cnt <= kWaveKeys;               // (more keys than a wave sorts: the walk is over)
            }
        }
        int nr0, nr1;
        box_rows(p_nxt, is_mine(s_nxt), nr0, nr1);                  // (requested here, used by the next turn of the loop)
        if (mine && (cands > kWaveCandidates || cnt > kWaveKeys)) {   // (uniform) not a list for one wave
            if (lane == 0) huge_list[atomicAdd(&ds->huge_count, 1)] = s;
        } else if (mine) {
            if (chunk_end - chunk_pos < (unsigned long long)cnt) {           // (uniform) a new chunk
                const unsigned long long take = (unsigned)cnt > chunk_keys ? (unsigned long long)cnt : (unsigned long long)chunk_keys;
                unsigned off_lo = 0u, off_hi = 0u;
                if (lane == 0) {
                    const unsigned long long o = atomicAdd(&ds->key_cursor, take);
                    off_lo = (unsigned)o;
                    off_hi = (unsigned)(o >> 32);
                }
                chunk_pos = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)off_hi) << 32) |
                            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)off_lo);
                chunk_end = chunk_pos + take;
            }
            const bool fits = chunk_pos + (unsigned long long)cnt <= v.key_cap;
            const unsigned long long off = fits ? chunk_pos : ~0ull;
            chunk_pos += (unsigned long long)cnt;
            if (lane == 0) {
                if (!fits) atomicCAS(&ds->status, kStatusOk, kStatusKeyCapacity);
                v.seg_start[s] = (unsigned)off;
                v.seg_len[s] = fits ? cnt : 0;
            }
            if (off != ~0ull) {                 // (no room: this call fails, the next one has it -- kpl_sync_status)
                unsigned long long *out = v.sort_keys + off;
                wave_lds_fence();               // the list is complete
                if (cnt <= kWave) wave_sort_store<1>(list, cnt, lane, r2, out);
                else if (cnt <= 2 * kWave) wave_sort_store<2>(list, cnt, lane, r2, out);
                else if (cnt <= 4 * kWave) wave_sort_store<4>(list, cnt, lane, r2, out);
                else wave_sort_store<8>(list, cnt, lane, r2, out);
            }
        }
        li += nwaves;
        s_cur = s_nxt;
        s_nxt = s_nn;
        p = p_nxt;
        my_r0 = nr0;
        my_r1 = nr1;
    }
}

__global__ __launch_bounds__(kCollectThreads) void sorted_collect_kernel(Batch b) {
    __shared__ unsigned long long keys[kCollectKeys];
    __shared__ int hist[kCollectBuckets];
    __shared__ int wsum[kCollectThreads / kWave];
    __shared__ int s_cnt, s_fallback, s_r0[16], s_r1[16];
    __shared__ unsigned long long s_off;
    const ViewDev &v = b.view[blockIdx.y];
    if (!v.f.sorted) return;
    DevState *ds = v.ds;
    const int nlarge = ds->huge_count;                      // what sorted_collect_wave_kernel left to this kernel
    const int *__restrict__ huge_list = v.large_list + v.n;
    const GridDesc g = ds->grid;
    const float4 *__restrict__ pts = v.pts;
    const int *__restrict__ cell_start = v.cell_start;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    const float r2 = v.f.r2;
    const unsigned long long key_end = (unsigned long long)__float_as_uint(r2) << 32;     // every key is below it (d2 < r2)
    for (int li = blockIdx.x; li < nlarge; li += gridDim.x) {
        const int s = huge_list[li];
        const float4 p = pts[s];
        CellBox bx = make_box(g, p.x, p.y, p.z, v.f.rr);
        bx.hi[1] = min(bx.hi[1], bx.lo[1] + 3);
        bx.hi[2] = min(bx.hi[2], bx.lo[2] + 3);
        // the rows of the box once per point: first position and end of each (thread r < 16: row r; empty where the box
        // has none) -- the walk below reads them from LDS instead of waiting for cell_start[] row after row
        __syncthreads();
        if (tid < 16) {
            const int z = bx.lo[2] + tid / 4, y = bx.lo[1] + tid % 4;
            const bool valid = z <= bx.hi[2] && y <= bx.hi[1];
            const int row = valid ? (z * g.dims[1] + y) * g.dims[0] : 0;
            s_r0[tid] = valid ? cell_start[row + bx.lo[0]] : 0;
            s_r1[tid] = valid ? cell_start[row + bx.hi[0] + 1] : 0;
        }
        __syncthreads();
        // the keys of the window [lo, hi) -> keys[] (the first kCollectKeys of them); returns how many there are.  The rows are
        // walked in pieces of 1 024 consecutive candidates (4 per thread, coalesced 16-byte loads); the loads of the next
        // piece are in flight while the current one is tested
        auto collect = [&](unsigned long long lo, unsigned long long hi) -> int {
            __syncthreads();
            if (tid == 0) s_cnt = 0;
            __syncthreads();
            constexpr int kPiece = kCollectThreads * kCollectAhead;
            const bool whole = lo == 0ull && hi == key_end;  // (the usual call: no window to test the keys against)
            int row = -1, t0 = 0, r1 = 0;                    // (uniform) the piece [t0, min(t0 + kPiece, r1)) of row `row`
            auto advance = [&]() -> bool {
                t0 += kPiece;
                while (t0 >= r1) {
                    if (++row >= 16) return false;
                    t0 = s_r0[row];
                    r1 = s_r1[row];
                }
                return true;
            };
            auto load_piece = [&](float4 (&q)[kCollectAhead]) {
#pragma unroll
                for (int a = 0; a < kCollectAhead; ++a) q[a] = pts[min(t0 + a * kCollectThreads + tid, r1 - 1)];
            };
            float4 cur[kCollectAhead], nxt[kCollectAhead];
            bool have = advance();
            if (have) load_piece(cur);
            while (have) {
                const int c_t0 = t0, c_r1 = r1;
                const bool have_next = advance();
                if (have_next) load_piece(nxt);
                unsigned long long key[kCollectAhead], bal[kCollectAhead];
                bool take[kCollectAhead];
                int mine = 0;
#pragma unroll
                for (int a = 0; a < kCollectAhead; ++a) {
                    const float d2 = dist2(p.x, p.y, p.z, cur[a]);
                    key[a] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(unsigned)__float_as_int(cur[a].w);
                    take[a] = c_t0 + a * kCollectThreads + tid < c_r1 && d2 < r2;                  // strict (KdTreeFLANN)
                    if (!whole) take[a] = take[a] && key[a] >= lo && key[a] < hi;
                    bal[a] = __ballot(take[a]);
                    mine += __popcll(bal[a]);
                }
                int base = 0;
                if (lane == 0 && mine > 0) base = atomicAdd(&s_cnt, mine);                // one LDS atomic per wave and piece
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int a = 0; a < kCollectAhead; ++a) {
                    const int slot = base + __popcll(bal[a] & ((1ull << lane) - 1ull));
                    if (take[a] && slot < kCollectKeys) keys[slot] = key[a];
                    base += __popcll(bal[a]);
                }
#pragma unroll
                for (int a = 0; a < kCollectAhead; ++a) cur[a] = nxt[a];
                have = have_next;
            }
            __syncthreads();
            return s_cnt;
        };
        // ascending order of keys[0 .. cnt), cnt <= kCollectKeys; the d2 of the keys lie in [lo_d2, hi_d2)
        auto sort_list = [&](int cnt, float lo_d2, float hi_d2) {
            constexpr int kPer = kCollectKeys / kCollectThreads;                  // list elements per thread
            constexpr int kBPer = kCollectBuckets / kCollectThreads;              // buckets per thread
            const float span = hi_d2 - lo_d2;
            const float scale = span > 0.0f ? (float)kCollectBuckets / span : 0.0f;
            // bucket of a key: monotonic in d2 (float subtract, multiply by a positive constant, truncate, clamp)
            auto bucket_of = [&](unsigned long long k) -> int {
                const float t = (__uint_as_float((unsigned)(k >> 32)) - lo_d2) * scale;
                return min(max((int)t, 0), kCollectBuckets - 1);
            };
            for (int k = tid; k < kCollectBuckets; k += kCollectThreads) hist[k] = 0;
            if (tid == 0) s_fallback = 0;
            __syncthreads();
            const int nper = (cnt + kCollectThreads - 1) / kCollectThreads;      // (uniform) rounds of the list that hold keys
            unsigned long long mykey[kPer];
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                mykey[j] = ~0ull;
                if (j < nper) {
                    const int i = tid + j * kCollectThreads;
                    if (i < cnt) {
                        mykey[j] = keys[i];
                        atomicAdd(&hist[bucket_of(mykey[j])], 1);
                    }
                }
            }
            __syncthreads();
            // exclusive scan of the bucket counts: kBPer consecutive buckets per thread
            int local[kBPer], sum = 0;
#pragma unroll
            for (int k = 0; k < kBPer; ++k) {
                local[k] = hist[tid * kBPer + k];
                sum += local[k];
            }
            int incl = sum;
            for (int off = 1; off < kWave; off <<= 1) {
                const int o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            if (lane == kWave - 1) wsum[wid] = incl;
            __syncthreads();
            int run = incl - sum;
            for (int w = 0; w < wid; ++w) run += wsum[w];
#pragma unroll
            for (int k = 0; k < kBPer; ++k) {
                hist[tid * kBPer + k] = run;                                      // the bucket's cursor = its first position
                run += local[k];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kPer; ++j)                                      // (every key is in a register: the list may be overwritten)
                if (j < nper && tid + j * kCollectThreads < cnt) keys[atomicAdd(&hist[bucket_of(mykey[j])], 1)] = mykey[j];
            __syncthreads();
            // every thread orders the keys of its kBPer consecutive buckets (one contiguous piece of the list; a cursor has
            // come to rest at the end of its bucket = the start of the next)
            const int a0 = tid == 0 ? 0 : hist[tid * kBPer - 1], a1 = hist[tid * kBPer + kBPer - 1];
            if (a1 - a0 > kInsertionMax) s_fallback = 1;
            else
                for (int i = a0 + 1; i < a1; ++i) {
                    const unsigned long long x = keys[i];
                    int j = i - 1;
                    while (j >= a0 && keys[j] > x) {
                        keys[j + 1] = keys[j];
                        --j;
                    }
                    keys[j + 1] = x;
                }
            __syncthreads();
            if (s_fallback) {                   // many equal or clustered distances: a bitonic network over the whole list
                int N = 2;
                while (N < cnt) N <<= 1;
                for (int i = cnt + tid; i < N; i += kCollectThreads) keys[i] = ~0ull;
                __syncthreads();
                for (int k = 2; k <= N; k <<= 1)
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = tid; i < N; i += kCollectThreads) {
                            const int ixj = i ^ j;
                            if (ixj > i) {
                                const unsigned long long x = keys[i], y = keys[ixj];
                                const bool up = (i & k) == 0;
                                if ((x > y) == up) {
                                    keys[i] = y;
                                    keys[ixj] = x;
                                }
                            }
                        }
                        __syncthreads();
                    }
            }
        };
        // ---- the whole neighborhood, or its first kCollectKeys keys and their number
        const int total = collect(0ull, key_end);
        if (tid == 0) {
            const unsigned long long off = atomicAdd(&ds->key_cursor, (unsigned long long)total);
            const bool fits = off + (unsigned long long)total <= v.key_cap;
            if (!fits) atomicCAS(&ds->status, kStatusOk, kStatusKeyCapacity);
            v.seg_start[s] = (unsigned)off;
            v.seg_len[s] = fits ? total : 0;
            s_off = fits ? off : ~0ull;
        }
        __syncthreads();
        const unsigned long long off = s_off;
        if (off == ~0ull) continue;                              // no room: this call fails, the next one has it (kpl_sync_status)
        unsigned long long *out = v.sort_keys + off;
        if (total <= kCollectKeys) {
            sort_list(total, 0.0f, r2);
            for (int i = tid; i < total; i += kCollectThreads) out[i] = keys[i];
            continue;
        }
        // ---- more keys than the list holds: windows of d2, sized for an even spread of the neighbors over d2 and halved
        // when one runs over (as point_features_sorted does)
        unsigned long long lo = 0ull;
        int done = 0;
        const float step = r2 * ((float)(kCollectKeys * 3 / 4) / (float)total);
        while (lo < key_end) {
            const float lo_d2 = __uint_as_float((unsigned)(lo >> 32));
            unsigned hb = __float_as_uint(lo_d2 + step);
            if (hb <= (unsigned)(lo >> 32)) hb = (unsigned)(lo >> 32) + 1u;
            unsigned long long hi = (unsigned long long)hb << 32;
            if (hi > key_end) hi = key_end;
            int cnt;
            for (;;) {
                cnt = collect(lo, hi);
                if (cnt <= kCollectKeys) break;
                const unsigned lb = (unsigned)(lo >> 32), hbb = (unsigned)(hi >> 32);
                if (hbb - lb >= 2u) {
                    const float ld = __uint_as_float(lb), hd = __uint_as_float(hbb);
                    unsigned mb = __float_as_uint(ld + (hd - ld) * 0.5f);
                    if (mb <= lb || mb >= hbb) mb = lb + (hbb - lb) / 2u;
                    hi = (unsigned long long)mb << 32;
                } else {
                    hi = lo + (hi - lo) / 2ull;                  // one or two distances left: halve by the whole key
                }
            }
            // (the keys of a window whose upper end is not a whole d2 start in the middle of one: the bucket range covers it)
            const float w_lo = __uint_as_float((unsigned)(lo >> 32));
            const float w_hi = (hi & 0xffffffffull) ? __uint_as_float((unsigned)(hi >> 32) + 1u) : __uint_as_float((unsigned)(hi >> 32));
            sort_list(cnt, w_lo, fmaxf(w_hi, w_lo));
            for (int i = tid; i < cnt; i += kCollectThreads) out[done + i] = keys[i];
            done += cnt;
            lo = hi;
        }
    }
}

// the feature loop (hpp:334-359) of the large points over their sorted segments; writes their columns of the feature block.
// Persistent waves over ViewDev::large_list, G lanes per point: two when the large points alone fill the chip, four when
// they are few (a 63 k-point view with 2 300 neighbors per point is 2 000 waves of two lanes per point -- two per SIMD,
// each alive for the whole kernel; four lanes per point are twice the waves and half the rounds per wave).
constexpr int kAddWideBelow = 160 * 1024;       // large points of a view below which four lanes take a point
constexpr unsigned long long kAddWideKeys = 1024;       // ... or mean keys per point below which they do: lists of a few hundred
                                                        // keys are over in ~50 rounds of four lanes (8 x 200 k points at 200 / 290
                                                        // keys: 8.2 -> 7.6, 11.6 -> 10.7 ms for the stage)

template <bool STATS, int G>
__device__ __forceinline__ void sorted_add_points(const ViewDev &v, float *H, int nlarge, int bx) {
    constexpr int kPts = kLanes / G;
    const int pi = threadIdx.x / G, gq = threadIdx.x % G;
    FeatDesc f;
    f.A = pin_i(v.f.A);
    f.B = pin_i(v.f.B);
    f.F = pin_i(v.f.F);
    f.A1f = pin_f(v.f.A1f);
    f.B1f = pin_f(v.f.B1f);
    f.support = v.f.support;
    f.ann_dim = pin_f(v.f.ann_dim);
    f.ann_half = pin_f(v.f.ann_half);
    f.ann_rdim = pin_f(v.f.ann_rdim);
    f.bin_dim = pin_f(v.f.bin_dim);
    f.bin_half = pin_f(v.f.bin_half);
    f.bin_rdim = pin_f(v.f.bin_rdim);
    f.r2 = pin_f(v.f.r2);
    f.rr = v.f.rr;
    const char *__restrict__ nrmsrc = v.nrmsrc;
    const unsigned ns = v.ns;
    const int col_address = lds_address(H + pi);
    for (int first = bx * kPts; first < nlarge; first += gridDim.x * kPts) {
        const bool has_point = first + pi < nlarge;
        const int s = has_point ? (int)((unsigned)v.large_list[first + pi] & ~kHugeBit) : 0;
        wave_lds_fence();
        for (int c = gq; c < f.F; c += G) H[c * kPts + pi] = 0.0f;                 // hpp:325
        wave_lds_fence();
        const int len = has_point ? v.seg_len[s] : 0;
        const unsigned long long *seg = v.sort_keys + (has_point ? v.seg_start[s] : 0u);
        const float4 np = has_point ? v.nrm[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (STATS && has_point && gq == 0) atomicAdd(&v.stats->sum_kf, (unsigned long long)len);
        // element 0 of the order is dropped (hpp:336).  Keys are requested FOUR rounds ahead of their use and normals TWO: the
        // address of a normal comes out of a key, and with the key one round ahead (until r04e) every round waited for a
        // whole memory round trip -- 5.8 k cycles per round of 79 instructions at 200 keys per point.
        struct Slot {
            bool valid;
            float d2;
            unsigned orig;
            f32x3 n;
        };
        int k = 1 + gq;
        auto fetch_key = [&](Slot &slot) {
            slot.valid = k < len;
            const unsigned long long key = seg[slot.valid ? k : 0];
            k += G;
            slot.d2 = __uint_as_float((unsigned)(key >> 32));
            slot.orig = slot.valid ? (unsigned)key : 0u;
        };
        auto fetch_normal = [&](Slot &slot) { slot.n = *reinterpret_cast<const f32x3 *>(nrmsrc + (size_t)slot.orig * ns); };
        constexpr int kSlots = 6, kKeyAhead = 4, kNormalAhead = 2;
        Slot sl[kSlots];
#pragma unroll
        for (int q = 0; q < kSlots; ++q) {
            sl[q].valid = false;
            sl[q].d2 = 0.f;
            sl[q].orig = 0u;
            sl[q].n = f32x3{0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < kKeyAhead; ++q) fetch_key(sl[q]);
#pragma unroll
        for (int q = 0; q < kNormalAhead; ++q) fetch_normal(sl[q]);
        // (slots become invalid in key order: a group is done when its current slot is)
        bool more = true;
        while (more) {
#pragma unroll
            for (int q = 0; q < kSlots; ++q) {
                Slot &now = sl[q];
                if (!__any(now.valid)) {
                    more = false;
                    break;
                }
                fetch_key(sl[(q + kKeyAhead) % kSlots]);
                fetch_normal(sl[(q + kNormalAhead) % kSlots]);
                const bool has_ = now.valid & finite3(now.n.x, now.n.y, now.n.z);          /* hpp:338 */
                Contribution c_;
                if (has_) c_ = neighbor_contribution<kPts>(f, now.d2, np, now.n, col_address);
#pragma unroll
                for (int sub_ = 0; sub_ < G; ++sub_) {
                    if (has_ & (gq == sub_)) apply_contribution(c_, request_cells(c_));
                    wave_lds_fence();
                }
            }
        }
        wave_lds_fence();
        for (int a = gq; a < f.A; a += G) {                                        // hpp:360-370, one row per lane
            float *h = H + (a * f.B) * kPts + pi;
            float ssum = 0.0f;
            for (int kk = 0; kk < f.B; ++kk) {
                float x = h[kk * kPts];
                ssum += x * x;
            }
            const float nr = sqrtf(ssum);
            if (nr > 0)
                for (int kk = 0; kk < f.B; ++kk) h[kk * kPts] = h[kk * kPts] / nr;
        }
        wave_lds_fence();
        if (has_point) {
            float *o = v.feat + (size_t)(s / kLanes) * f.F * kLanes + (s % kLanes);
            for (int c = gq; c < f.F; c += G) o[c * kLanes] = H[c * kPts + pi];
        }
    }
}

template <bool STATS>
__global__ __launch_bounds__(kLanes) void sorted_add_kernel(Batch b, int maxF, int by_xcd) {
    extern __shared__ float H[];
    const ViewBlock vb = view_block(by_xcd);
    const ViewDev &v = b.view[vb.view];
    if (!v.f.sorted) return;
    const int nlarge = v.ds->large_count;
    if (nlarge == 0) return;
    if (v.ds->status != kStatusOk) return;                        // (segments incomplete: the call fails, kpl_sync_status)
    // (all segments have been handed out: the cursor is the number of keys of the view, chunk tails included)
    const bool wide = nlarge < kAddWideBelow || v.ds->key_cursor < (unsigned long long)nlarge * kAddWideKeys;
    if (wide) sorted_add_points<STATS, 4>(v, H, nlarge, (int)vb.bx);
    else sorted_add_points<STATS, 2>(v, H, nlarge, (int)vb.bx);
}

// detectKeypoints, hpp:197-256 with draws_remove == false (order-independent predicate):
// keypoint <=> score >= thr (float promoted to double, hpp:207 -- tested by the forest kernel,
// which appends the candidates to `cand`) and no neighbor within r_nms has a strictly greater
// score (hpp:219).  L lanes share one candidate and walk the rows of cells of its search box one after
// the other, kNmsAhead positions per lane and step, scores first: a position only matters if its score is
// greater (or, for the draws pass, equal) -- only then is its point loaded and its distance tested --, and the
// group stops at the first greater neighbor inside the radius (most candidates are not maxima).
// L is a choice between latency and work (launch_post): for one view alone on the GPU the kernel is a chain of
// dependent loads -- 16 lanes, 8 when the view has so many candidates that the groups of 16 would need several rounds
// (62 k points 0.020 ms; 500 k points at r_nms = 4 mr 0.030 ms); for a batch, where the other batch's kernels fill the GPU
// anyway, what counts is the number of instructions -- 4 lanes (8 views of 200 k points, two batches in flight: 1 888 ->
// 1 940 Mpoints/s; one lane per candidate: 1 946, but 64 views of 63 k points 1 380 instead of 1 445).  Until r03f
// one kernel took 16 lanes per candidate and treated the rows as one list (prefix sums + a binary search per position
// through ds_bpermute): 75 wave-instructions per candidate (profiles/r03_notes.md).
constexpr int kNmsAhead = 4;     // positions per lane whose score loads are in flight together

template <bool STATS, int L>
__device__ __forceinline__ void nms_scan(const ViewDev &v) {
    const float4 *__restrict__ pts = v.pts;
    const int *__restrict__ cell_start = v.cell_start;
    const NmsDesc nd = v.nd;
    if (!nd.non_maxima) return;
    const float *__restrict__ score_sorted = v.score_sorted;
    const NmsList cand = v.cand;
    int *__restrict__ flags = v.flags;
    StatsDev *stats = v.stats;
    const bool DRAWS = nd.draws_remove != 0;
    const GridDesc g = v.ds->grid;
    const int ncand = *cand.count;
    const int t_last = max(cell_start[g.ncells] - 1, 0);
    const int lane = threadIdx.x & (L - 1), gbase = (threadIdx.x & (kWave - 1)) & ~(L - 1);
    const unsigned long long gmask = (1ull << L) - 1ull;
    // (the lanes of a group take the same branches: a ballot inside these loops sees the whole group)
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) / L; k < ncand; k += gridDim.x * (blockDim.x / L)) {
        const int s = cand.list[k];
        const float si = score_sorted[s];
        const float4 p = pts[s];
        const CellBox bx = make_box(g, p.x, p.y, p.z, nd.rr);
        bool greater = false, draw = false;
        int kn = 0;
        for (int cz = bx.lo[2]; cz <= bx.hi[2] && (STATS || !greater); ++cz)
            for (int cy = bx.lo[1]; cy <= bx.hi[1] && (STATS || !greater); ++cy) {
                const int row = (cz * g.dims[1] + cy) * g.dims[0];
                const int r0 = cell_start[row + bx.lo[0]], r1 = cell_start[row + bx.hi[0] + 1];
                for (int base = r0; base < r1 && (STATS || !greater); base += L * kNmsAhead) {
                    const int t0 = base + lane * kNmsAhead;
                    float sj[kNmsAhead];
#pragma unroll
                    for (int a = 0; a < kNmsAhead; ++a) sj[a] = score_sorted[min(t0 + a, t_last)];
                    bool hit = false, tie = false;
#pragma unroll
                    for (int a = 0; a < kNmsAhead; ++a) {
                        const int t = t0 + a;
                        const bool in = t < r1;
                        const bool gt = in && si < sj[a];                                  // hpp:219
                        const bool eq = DRAWS && in && si == sj[a] && t != s;              // hpp:224-229
                        if (STATS ? in : (gt || eq)) {
                            const float4 q = pts[t];
                            const bool inside = dist2(p.x, p.y, p.z, q) < nd.r2;
                            if (STATS) kn += inside;
                            hit |= inside && gt;
                            tie |= inside && eq;
                        }
                    }
                    if (L == 1) {
                        greater |= hit;
                        draw |= tie;
                    } else {
                        greater |= ((__ballot(hit) >> gbase) & gmask) != 0ull;
                        if (DRAWS) draw |= ((__ballot(tie) >> gbase) & gmask) != 0ull;
                    }
                }
            }
        if (STATS) {
            atomicAdd(&stats->sum_kn, (unsigned long long)kn);
            if (lane == 0) atomicAdd(&stats->n_thresholded, 1ull);
        }
        // 1 = keypoint (hpp:252-253); 2 = maximum with draws, decided by the draws pass (hpp:233-250)
        if (lane == 0 && !greater) flags[__float_as_int(p.w)] = (DRAWS && draw) ? 2 : 1;
    }
}

// L lanes per candidate, or LMANY when the view has more than `many` candidates (0 = never)
template <bool STATS, int L, int LMANY>
__global__ __launch_bounds__(256) void nms_kernel(Batch b, int many) {
    const ViewDev &v = b.view[blockIdx.y];
    if (LMANY != 0 && v.nd.non_maxima && *v.cand.count > many) nms_scan<STATS, LMANY == 0 ? L : LMANY>(v);
    else nms_scan<STATS, L>(v);
}

// non_maxima_draws_remove == true, hpp:231-250: the order-dependent greedy pass over the maxima that have equal-score
// neighbors ("draws"), in ascending point index.  The reference walks them in order: a maximum on its skip list is
// dropped; another one survives iff some draw lies within draws_threshold, and every such draw goes on the skip list.
// So a listed maximum i is dropped iff a listed maximum j < i that is NOT dropped has i within the threshold -- the
// lexicographically first maximal independent set of the "draw within the threshold" graph, plus the rule that a
// maximum without any such draw (listed or not) does not survive either.  That can be evaluated out of order: an
// entry is decided as soon as every lower-index neighbor of its is (states in skip[]: 3 undecided, 1 kept its place, 2
// dropped; 0 = not a listed maximum).  kDrawRounds parallel rounds (16 lanes per entry, entries that are still waiting
// counted per round: a round whose predecessor left nothing returns at once), then one wave takes what is left in index
// order -- long chains of maxima that each wait for the one before (points in scan order along a plateau).  One wave
// over the whole list took 25 ms for the 22 000 listed maxima of a 200 k-point view; now 0.3-0.6 ms.
constexpr int kDrawKept = 1, kDrawDropped = 2, kDrawUndecided = 3;
// draw_list: [n] listed maxima, [n] adjacency counts, then -- from a 16-byte boundary -- [n x kDrawAdj] adjacency rows
__host__ __device__ inline size_t draw_adj_offset(int n) { return 2 * (((size_t)(n > 0 ? n : 1) + 3) & ~(size_t)3); }

// one listed maximum, LANES lanes: does any draw lie within the threshold, is a lower-index neighbor kept / undecided
struct SkipStates {           // the states as the parallel rounds see them: skip[] in memory
    const int *skip;
    __device__ __forceinline__ int operator()(int qi) const { return __hip_atomic_load(&skip[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
template <int LANES, class States>
__device__ __forceinline__ void draws_sweep(const ViewDev &v, const GridDesc &g, int idx, int lane, bool &any_draw,
                                            bool &lower_kept, bool &lower_undecided, const States &state_of_point) {
    const NmsDesc nd = v.nd;
    const float4 *__restrict__ pts = v.pts;
    const int *__restrict__ cell_start = v.cell_start;
    const float *__restrict__ score_sorted = v.score_sorted;
    const int s = v.pos_of[idx];
    const float4 p = pts[s];
    const float si = score_sorted[s];
    const CellBox b = make_box(g, p.x, p.y, p.z, nd.rr);
    any_draw = lower_kept = lower_undecided = false;
    for (int cz = b.lo[2]; cz <= b.hi[2]; ++cz)
        for (int cy = b.lo[1]; cy <= b.hi[1]; ++cy) {
            const int row = (cz * g.dims[1] + cy) * g.dims[0];
            const int t0 = cell_start[row + b.lo[0]], t1 = cell_start[row + b.hi[0] + 1];
            for (int t = t0 + lane; t < t1; t += LANES) {
                const float4 q = pts[t];
                if (t != s && dist2(p.x, p.y, p.z, q) < nd.r2 && si == score_sorted[t]) {
                    // hpp:239 (a - b).norm(): x*x + (y*y + z*z), then sqrt
                    const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                    const float distance = sqrtf(dx * dx + (dy * dy + dz * dz));
                    if (distance < nd.draws_thr) {                                  // hpp:240
                        any_draw = true;
                        const int qi = __float_as_int(q.w);
                        if (qi < idx) {                                             // it comes first in the reference's loop
                            const int sq = state_of_point(qi);
                            lower_kept |= sq == kDrawKept;
                            lower_undecided |= sq == kDrawUndecided;
                        }
                    }
                }
            }
        }
}

constexpr int kDrawLanes = 16;

// The pass in three steps.  draws_adj_kernel, parallel, pure geometry -- the ONE sweep of a listed maximum's neighborhood:
// has it a draw within the threshold at all, and which listed maxima of lower index lie within it; their LIST POSITIONS go to
// the entry's adjacency row (at most kDrawAdj; the count runs on beyond that).  An entry without such a neighbor is decided
// on the spot.  No state is read: the rows are the same whatever the order the groups run in.
__global__ __launch_bounds__(256) void draws_adj_kernel(Batch b) {
    const ViewDev &v = b.view[blockIdx.y];
    if (!v.nd.draws_remove) return;
    DevState *ds = v.ds;
    if (ds->draws_left[0] == 0) return;                      // nothing listed
    const NmsDesc nd = v.nd;
    const float4 *__restrict__ pts = v.pts;
    const int *__restrict__ cell_start = v.cell_start;
    const float *__restrict__ score_sorted = v.score_sorted;
    const int count = *v.draw_count;
    int *adjn = v.draw_list + v.n, *adj = v.draw_list + draw_adj_offset(v.n);
    const GridDesc g = ds->grid;
    const int lane = threadIdx.x & (kDrawLanes - 1), gbase = (threadIdx.x & (kWave - 1)) & ~(kDrawLanes - 1);
    const unsigned long long gmask = (1ull << kDrawLanes) - 1ull;
    auto group_any = [&](bool x) { return ((__ballot(x) >> gbase) & gmask) != 0ull; };
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) / kDrawLanes; k < count; k += gridDim.x * (blockDim.x / kDrawLanes)) {
        const int idx = v.draw_list[k];
        const int s = v.pos_of[idx];
        const float4 p = pts[s];
        const float si = score_sorted[s];
        const CellBox bx = make_box(g, p.x, p.y, p.z, nd.rr);
        bool any_draw = false, lower = false;
        for (int cz = bx.lo[2]; cz <= bx.hi[2]; ++cz)
            for (int cy = bx.lo[1]; cy <= bx.hi[1]; ++cy) {
                const int row = (cz * g.dims[1] + cy) * g.dims[0];
                const int t0 = cell_start[row + bx.lo[0]], t1 = cell_start[row + bx.hi[0] + 1];
                for (int t = t0 + lane; t < t1; t += kDrawLanes) {
                    const float4 q = pts[t];
                    if (t != s && dist2(p.x, p.y, p.z, q) < nd.r2 && si == score_sorted[t]) {
                        const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;          // hpp:239, as in draws_sweep
                        const float distance = sqrtf(dx * dx + (dy * dy + dz * dz));
                        if (distance < nd.draws_thr) {                                      // hpp:240
                            any_draw = true;
                            const int qi = __float_as_int(q.w);
                            // a LISTED maximum of lower index (skip[] != 0 since list_match_kernel; decisions of this
                            // launch change 3 into 1 or 2, never into 0)
                            if (qi < idx && __hip_atomic_load(&v.skip[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                                lower = true;
                                const int slot = atomicAdd(&adjn[k], 1);
                                if (slot < kDrawAdj) adj[(size_t)k * kDrawAdj + slot] = v.prefix[qi];   // its position in the list
                            }
                        }
                    }
                }
            }
        any_draw = group_any(any_draw);
        lower = group_any(lower);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the group's appends, before lane 0 tags the count
        if (lane != 0) continue;
        if (!lower) {                                                               // hpp:245-248: nobody before it can strike it
            v.flags[idx] = any_draw ? 1 : 0;
            __hip_atomic_store(&v.skip[idx], any_draw ? kDrawKept : kDrawDropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (any_draw) atomicOr(&adjn[k], 1 << 30);
            atomicAdd(&ds->draws_left[1], 1);
        }
    }
}

// kDrawRounds parallel rounds ON THE ROWS (no geometry): an undecided entry reads the states of the listed neighbors before
// it -- dropped if one of them is kept (hpp:234), decided if none is undecided any more (hpp:245-248), else it waits.  With
// the points in arbitrary order a handful of rounds decide nearly everything (the dependency chains are short); in scan
// order every round peels one front off the plateau and the rest is draws_rest_kernel's.  A round whose predecessor left
// nothing returns at once.  An entry with more than kDrawAdj neighbors sweeps its neighborhood instead (draws_sweep).
// (Until r04a these rounds swept the neighborhood of every undecided entry again: 8 x 75 us per 200 k-point view in scan
// order, against 8 x 6 us.)
__global__ __launch_bounds__(256) void draws_round_kernel(Batch b, int round) {
    const ViewDev &v = b.view[blockIdx.y];
    if (!v.nd.draws_remove) return;
    DevState *ds = v.ds;
    if (ds->draws_left[round + 1] == 0) return;              // nothing listed, or everything decided by the rounds before
    const int count = *v.draw_count;
    const int *adjn = v.draw_list + v.n, *adj = v.draw_list + draw_adj_offset(v.n);
    const GridDesc g = ds->grid;
    const int lane = threadIdx.x & (kDrawLanes - 1), gbase = (threadIdx.x & (kWave - 1)) & ~(kDrawLanes - 1);
    const unsigned long long gmask = (1ull << kDrawLanes) - 1ull;
    auto group_any = [&](bool x) { return ((__ballot(x) >> gbase) & gmask) != 0ull; };
    for (int k = (blockIdx.x * blockDim.x + threadIdx.x) / kDrawLanes; k < count; k += gridDim.x * (blockDim.x / kDrawLanes)) {
        const int idx = v.draw_list[k];
        if (__hip_atomic_load(&v.skip[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kDrawUndecided) continue;
        const int nraw = adjn[k], nadj = nraw & 0xffffff;
        bool any_draw = ((nraw >> 30) & 1) != 0, lower_kept = false, lower_undecided = false;
        if (nadj > kDrawAdj) {
            bool a;
            draws_sweep<kDrawLanes>(v, g, idx, lane, a, lower_kept, lower_undecided, SkipStates{v.skip});
        } else {
            for (int q = lane; q < nadj; q += kDrawLanes) {
                const int sq = __hip_atomic_load(&v.skip[v.draw_list[adj[(size_t)k * kDrawAdj + q]]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lower_kept |= sq == kDrawKept;
                lower_undecided |= sq == kDrawUndecided;
            }
        }
        lower_kept = group_any(lower_kept);
        lower_undecided = group_any(lower_undecided);
        if (lane != 0) continue;
        if (lower_kept) {                                                           // hpp:234: on the skip list
            v.flags[idx] = 0;
            __hip_atomic_store(&v.skip[idx], kDrawDropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!lower_undecided) {                                              // hpp:245-248
            v.flags[idx] = any_draw ? 1 : 0;
            __hip_atomic_store(&v.skip[idx], any_draw ? kDrawKept : kDrawDropped, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            atomicAdd(&ds->draws_left[round + 2], 1);
        }
    }
}

// ... and draws_rest_kernel, ONE WORKGROUP per view, in index order: the states of all listed maxima as 2-bit fields in
// LDS, the list cut into chunks of 64 entries.  Chunk c belongs to wave c mod W.  What does not depend on earlier decisions
// -- the adjacency rows of the chunk, which neighbors lie inside it -- is prepared by the wave at once; then it waits until
// every chunk before its own is final (a counter in LDS, advanced chunk by chunk), reads the states of the neighbors before
// the chunk, settles the chunk on scalars and hands on.  Scan order makes this a pipeline: a maximum of a plateau waits for
// its left neighbor (the same chunk, or the one before) and for the row above (a chunk or two back), so the serial part
// per chunk is a few LDS reads and one scalar loop over the entries that are KEPT -- a kept entry strikes every entry of the
// chunk that waits for it from the candidates with one ballot, so an entry still standing when its turn comes is kept.
// One wave alone (round 3) spent its time fetching rows between the serial parts: 5 ms per 200 k-point view in scan order.
// An entry with more than kDrawAdj waiting neighbors sweeps its neighborhood again (draws_sweep, states from LDS through
// prefix[]: position in the list).  lds_words = state words that fit (16 entries each); a longer list falls back to ONE wave
// reading the states from skip[].
constexpr int kRestWaves = 8;
__global__ __launch_bounds__(kRestWaves *kWave) void draws_rest_kernel(Batch b, int lds_words) {
    extern __shared__ uint32_t dlds[];
    __shared__ int done;                                                 // chunks that are final
    const ViewDev &v = b.view[blockIdx.y];
    if (!v.nd.draws_remove) return;
    DevState *ds = v.ds;
    if (ds->draws_left[kDrawRounds + 1] == 0) return;
    const int count = *v.draw_count;
    const int *list = v.draw_list, *adjn = v.draw_list + v.n, *adj = v.draw_list + draw_adj_offset(v.n);
    const GridDesc g = ds->grid;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    const int nwords = (count + 15) / 16;
    const bool in_lds = nwords <= lds_words;
    uint32_t *state = dlds;
    int *rows = reinterpret_cast<int *>(dlds + lds_words) + wid * (kWave * kDrawAdj);      // [64][kDrawAdj] per wave
    if (in_lds) {
        for (int w = threadIdx.x; w < nwords; w += blockDim.x) {
            uint32_t word = 0u;
            for (int e = 0; e < 16; ++e) {
                const int k = w * 16 + e;
                if (k < count) word |= (uint32_t)(v.skip[list[k]] & 3) << (2 * e);
            }
            state[w] = word;
        }
    }
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    const int nwaves = in_lds ? kRestWaves : 1;                          // states in memory: one wave, its own stores in order
    if (wid >= nwaves) return;
    auto state_of = [&](int pos) -> int {
        if (in_lds) return (int)((state[pos >> 4] >> ((pos & 15) * 2)) & 3u);
        return __hip_atomic_load(&v.skip[list[pos]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // the state of POINT qi for a sweep: through its position in the list when the states live in LDS
    auto state_of_point = [&](int qi) -> int {
        if (!in_lds) return __hip_atomic_load(&v.skip[qi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int pos = v.prefix[qi];
        return (pos < count && list[pos] == qi) ? state_of(pos) : 0;
    };
    auto wait_for_turn = [&](int chunk) {                                // every chunk before this one is final
        if (nwaves == 1) return;
        while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < chunk) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
    };
    auto hand_on = [&](int chunk) {                                      // (a wave's LDS operations execute in order: the states first)
        if (nwaves == 1) return;
        wave_lds_fence();
        if (lane == 0) __hip_atomic_store(&done, chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const size_t adj_len = (size_t)v.n * kDrawAdj;
    const int nchunks = (count + 63) / 64;
    for (int chunk = wid; chunk < nchunks; chunk += nwaves) {
        const int k0 = chunk * 64;
        const int kk = k0 + lane;
        // (the entries of a chunk are decided by this wave only: their states do not change before its turn)
        const int my_state = kk < count ? state_of(kk) : kDrawDropped;
        unsigned long long todo = __ballot(my_state == kDrawUndecided);
        if (todo == 0ull) {
            wait_for_turn(chunk);
            hand_on(chunk);
            continue;
        }
        const int my_idx = kk < count ? list[kk] : 0, my_n = kk < count ? adjn[kk] : 0;
        const bool mine = my_state == kDrawUndecided;
        const int n_mine = my_n & 0xffffff;
        if (in_lds && !__any(mine && n_mine > kDrawAdj)) {
            // the usual chunk.  Before its turn: the lane's row straight from memory into registers (16 bytes per load, as
            // many groups of four as the longest row of the chunk needs), which neighbors lie inside the chunk, which before it
            constexpr int kGroups = kDrawAdj / 4;
            const int4 *row4 = reinterpret_cast<const int4 *>(adj + (size_t)min(kk, count - 1) * kDrawAdj);     // (16-byte aligned: draw_adj_offset)
            int longest = mine ? n_mine : 0;
            for (int off = kWave / 2; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off));
            const int ngroups = __builtin_amdgcn_readfirstlane((longest + 3) / 4);
            int before[kDrawAdj];
            unsigned long long inside = 0ull;
#pragma unroll
            for (int gi = 0; gi < kGroups; ++gi) {
                int4 r4 = make_int4(0, 0, 0, 0);
                if (gi < ngroups) r4 = row4[gi];
                const int rp[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool has = mine && 4 * gi + e < n_mine;
                    if (has && rp[e] >= k0) inside |= 1ull << (rp[e] - k0);
                    before[4 * gi + e] = (has && rp[e] < k0) ? rp[e] : -1;
                }
            }
            wait_for_turn(chunk);
            // its turn: the neighbors before the chunk are decided -- the state reads of a group in flight together
            bool kept_before = false;
#pragma unroll
            for (int gi = 0; gi < kGroups; ++gi)
                if (gi < ngroups) {
                    int st[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) st[e] = state_of(max(before[4 * gi + e], 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) kept_before |= before[4 * gi + e] >= 0 && st[e] == kDrawKept;
                }
            // (a neighbor inside the chunk may have been decided by its own group while the adjacency pass listed it: kept
            // already, it strikes the entries that wait for it here)
            unsigned long long kept_mask = __ballot(my_state == kDrawKept);
            const bool may_keep = mine && !kept_before && ((my_n >> 30) & 1) && (inside & kept_mask) == 0ull;
            const unsigned in_lo = (unsigned)inside, in_hi = (unsigned)(inside >> 32);
            // entries in ascending order: one that is still a candidate when its turn comes has no kept neighbor before it
            // in the chunk -- it is kept, and every entry that waits for it is struck
            unsigned long long cand = __ballot(may_keep);
            while (cand != 0ull) {
                const int j = __builtin_ctzll(cand);
                cand &= cand - 1ull;
                kept_mask |= 1ull << j;
                const unsigned wj = j < 32 ? in_lo : in_hi;
                cand &= ~__ballot(((wj >> (j & 31)) & 1u) != 0u);
            }
            const int final_state = mine ? (((kept_mask >> lane) & 1ull) ? kDrawKept : kDrawDropped) : my_state;
            const unsigned long long b0 = __ballot((final_state & 1) != 0), b1 = __ballot((final_state & 2) != 0);
            if (lane < 4 && (k0 >> 4) + lane < nwords) {                            // the chunk's four state words
                const unsigned lo = (unsigned)(b0 >> (16 * lane)) & 0xffffu, hi = (unsigned)(b1 >> (16 * lane)) & 0xffffu;
                uint32_t word = 0u;
                for (int e = 0; e < 16; ++e) word |= (((lo >> e) & 1u) | (((hi >> e) & 1u) << 1)) << (2 * e);
                state[(k0 >> 4) + lane] = word;
            }
            hand_on(chunk);
            if (mine) {                                                             // (nobody in this kernel reads these)
                v.flags[my_idx] = final_state == kDrawKept ? 1 : 0;                 // hpp:245-248
                __hip_atomic_store(&v.skip[my_idx], final_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        wave_lds_fence();
        for (int r = 0; r < kDrawAdj; ++r) {                             // the rows of the chunk, contiguous in memory
            const size_t at = (size_t)k0 * kDrawAdj + (size_t)r * 64 + lane;
            rows[r * 64 + lane] = at < adj_len ? adj[at] : 0;
        }
        wave_lds_fence();
        wait_for_turn(chunk);
        while (todo != 0ull) {                                           // a chunk with an entry that has to sweep: one by one
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int idx = __builtin_amdgcn_readlane(my_idx, j), nraw = __builtin_amdgcn_readlane(my_n, j);
            const int n = nraw & 0xffffff;
            bool any_draw = ((nraw >> 30) & 1) != 0, kept;
            if (n > kDrawAdj) {                                          // more neighbors than the row holds: sweep
                if (!in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (this wave's decisions must be in skip[])
                bool a, lk, lu;
                draws_sweep<64>(v, g, idx, lane, a, lk, lu, state_of_point);
                any_draw = __any(a);
                kept = __any(lk);
            } else {
                const int pos = lane < n ? rows[j * kDrawAdj + lane] : 0;
                kept = __any(lane < n && state_of(pos) == kDrawKept);
            }
            const int decision = (!kept && any_draw) ? kDrawKept : kDrawDropped;
            if (lane == 0) {
                if (in_lds) {
                    const int pos = k0 + j;
                    state[pos >> 4] = (state[pos >> 4] & ~(3u << ((pos & 15) * 2))) | ((uint32_t)decision << ((pos & 15) * 2));
                }
                v.flags[idx] = decision == kDrawKept ? 1 : 0;                       // hpp:245-248
                __hip_atomic_store(&v.skip[idx], decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!in_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_lds_fence();
        }
        hand_on(chunk);
    }
}

// computeCloudResolution, /root/reference/include/impl/point_cloud_utilities.hpp:120-151: for
// every point the square root of its second smallest squared distance (the smallest is the point
// itself), found by growing the searched block of cells until the answer is safe.  Values in
// ORIGINAL point order (NaN where the point is not finite or has no second neighbor).
__global__ __launch_bounds__(256) void second_nn_kernel(const float4 *__restrict__ pts,
                                                        const int *__restrict__ cell_start,
                                                        const int *__restrict__ pos_of,
                                                        const DevState *__restrict__ ds, int n,
                                                        float *__restrict__ val) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GridDesc g = ds->grid;
    const int s = g.ncells > 0 ? pos_of[i] : -1;
    if (s < 0) {
        val[i] = NAN;
        return;
    }
    const float4 p = pts[s];
    const int cx = cell_coord(p.x, g.mn[0], g.h, g.dims[0]);
    const int cy = cell_coord(p.y, g.mn[1], g.h, g.dims[1]);
    const int cz = cell_coord(p.z, g.mn[2], g.h, g.dims[2]);
    const int maxring = max(g.dims[0], max(g.dims[1], g.dims[2]));
    float best0 = INFINITY, best1 = INFINITY;
    for (int ring = 1; ring <= maxring; ++ring) {
        const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.dims[0] - 1);
        const int y0 = max(cy - ring, 0), y1 = min(cy + ring, g.dims[1] - 1);
        const int z0 = max(cz - ring, 0), z1 = min(cz + ring, g.dims[2] - 1);
        best0 = best1 = INFINITY;
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const int row = (z * g.dims[1] + y) * g.dims[0];
                const int t0 = cell_start[row + x0], t1 = cell_start[row + x1 + 1];
                for (int t = t0; t < t1; ++t) {
                    const float d2 = dist2(p.x, p.y, p.z, pts[t]);
                    if (d2 < best0) {
                        best1 = best0;
                        best0 = d2;
                    } else if (d2 < best1) {
                        best1 = d2;
                    }
                }
            }
        // every point outside the block is at least ring*h away (0.999: slack for the float cell edges)
        if (isfinite(best1) && (double)sqrtf(best1) <= (double)ring * (double)g.h * 0.999) break;
        if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dims[0] - 1 && y1 == g.dims[1] - 1 && z1 == g.dims[2] - 1) break;
    }
    val[i] = isfinite(best1) ? sqrtf(best1) : NAN;                                 // hpp:141
}

// res += sqrt(...) over the points in index order, double accumulator (hpp:141-148).  The sum of
// the reference is the SEQUENTIAL one.  All values are >= 0 and multiples of q = the smallest ulp
// among them; if the total stays below q * 2^52 every partial sum of every summation order is
// exactly representable, so a parallel reduction gives the sequential result bit for bit.  That
// is checked on the device (resolution_finish_kernel); only if it fails does one wave add the
// values one after the other.
struct SumPart {
    double sum;
    long long cnt;
    int emin;      // smallest ulp exponent among the non-zero values (INT_MAX if none)
    int pad;
};
constexpr int kSumBlocks = 256;

__device__ __forceinline__ SumPart sum_merge(SumPart a, const SumPart &b) {
    a.sum += b.sum;
    a.cnt += b.cnt;
    a.emin = min(a.emin, b.emin);
    return a;
}

__device__ __forceinline__ SumPart block_sum(SumPart v) {
    __shared__ SumPart red[256 / kWave];
    for (int off = kWave / 2; off > 0; off >>= 1) {
        SumPart o;
        o.sum = __shfl_xor(v.sum, off);
        o.cnt = __shfl_xor(v.cnt, off);
        o.emin = __shfl_xor(v.emin, off);
        v = sum_merge(v, o);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    SumPart t = red[0];
    for (int w = 1; w < 256 / kWave; ++w) t = sum_merge(t, red[w]);
    return t;
}

__global__ __launch_bounds__(256) void resolution_partial_kernel(const float *__restrict__ val, int n, SumPart *part) {
    SumPart v{0.0, 0, 0x7fffffff, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = val[i];
        if (!isnan(x)) {
            v.sum += (double)x;
            ++v.cnt;
            const int ef = (int)((__float_as_uint(x) >> 23) & 0xffu);
            if (x != 0.0f) v.emin = min(v.emin, max(ef, 1) - 150);
        }
    }
    v = block_sum(v);
    if (threadIdx.x == 0) part[blockIdx.x] = v;
}

// out[0] = sum, out[1] = count, out[2] = 1 if the parallel sum is the sequential one
__global__ __launch_bounds__(256) void resolution_finish_kernel(const SumPart *part, int nparts, double *out) {
    SumPart v{0.0, 0, 0x7fffffff, 0};
    if ((int)threadIdx.x < nparts) v = part[threadIdx.x];
    v = block_sum(v);
    if (threadIdx.x == 0) {
        const bool exact = v.emin == 0x7fffffff || v.sum < ldexp(1.0, v.emin + 52);
        out[0] = v.sum;
        out[1] = (double)v.cnt;
        out[2] = exact ? 1.0 : 0.0;
    }
}

// fallback: one wave, 64 values fetched at a time, added one after the other
__global__ __launch_bounds__(64) void ordered_sum_kernel(const float *__restrict__ val, int n, double *out) {
    if (out[2] != 0.0) return;
    double sum = 0.0;
    long long cnt = 0;
    for (int b = 0; b < n; b += 64) {
        const int i = b + threadIdx.x;
        const float v = i < n ? val[i] : NAN;
        for (int k = 0; k < 64; ++k) {
            const float vk = __shfl(v, k);
            if (!isnan(vk)) {
                sum += (double)vk;
                ++cnt;
            }
        }
    }
    if (threadIdx.x == 0) {
        out[0] = sum;
        out[1] = (double)cnt;
    }
}

// ---------------------------------------------------------------------------------------------
// Normal estimation: pcl::NormalEstimation as the reference uses it
// (/root/reference/src/main_test_detector.cpp:162-169 k-search 10, viewpoint (0,0,0);
// /root/reference/include/impl/KeypointLearning.hpp:125-148 radius search with search_radius_).
// Arithmetic as fixed in DESIGN.md section 2 (normal estimation): mean and covariance in
// double, two passes, sequential sums in neighbor order; cyclic Jacobi in double; flip towards the
// viewpoint.  One thread per ORIGINAL point; output in original order.
// ---------------------------------------------------------------------------------------------
struct NormalAcc {
    double sx, sy, sz;        // pass 1
    double mx, my, mz;        // mean
    double c00, c01, c02, c11, c12, c22;   // pass 2
    int m;
};

__device__ __forceinline__ void jacobi_smallest(double a[3][3], double v[3], double &lambda, double &trace) {
    double e[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 24; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double dg = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (!(off > 1e-36 * dg)) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[p][q];
                if (apq != 0.0) {
                    const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                    double t = 1.0 / (fabs(theta) + sqrt(theta * theta + 1.0));
                    if (theta < 0.0) t = -t;
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double akp = a[k][p], akq = a[k][q];
                        a[k][p] = c * akp - s * akq;
                        a[k][q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double apk = a[p][k], aqk = a[q][k];
                        a[p][k] = c * apk - s * aqk;
                        a[q][k] = s * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double ekp = e[k][p], ekq = e[k][q];
                        e[k][p] = c * ekp - s * ekq;
                        e[k][q] = s * ekp + c * ekq;
                    }
                }
            }
    }
    const bool m1 = a[1][1] < a[0][0];
    const double d01 = m1 ? a[1][1] : a[0][0];
    const bool m2 = a[2][2] < d01;
    lambda = m2 ? a[2][2] : d01;
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = m2 ? e[k][2] : (m1 ? e[k][1] : e[k][0]);
    trace = a[0][0] + a[1][1] + a[2][2];
}

__device__ __forceinline__ void store_normal(const NormalAcc &acc, const float4 &p, float vx, float vy,
                                             float vz, char *out, size_t out_stride, char *curv,
                                             size_t curv_stride, int i) {
    float *o = reinterpret_cast<float *>(out + (size_t)i * out_stride);
    float *cv = curv ? reinterpret_cast<float *>(curv + (size_t)i * curv_stride) : nullptr;
    if (acc.m < 3) {
        o[0] = o[1] = o[2] = NAN;
        if (cv) *cv = NAN;
        return;
    }
    double a[3][3] = {{acc.c00, acc.c01, acc.c02}, {acc.c01, acc.c11, acc.c12}, {acc.c02, acc.c12, acc.c22}};
    double v[3], lambda, trace;
    jacobi_smallest(a, v, lambda, trace);
    const double dot = v[0] * ((double)vx - (double)p.x) + v[1] * ((double)vy - (double)p.y) +
                       v[2] * ((double)vz - (double)p.z);
    if (dot < 0.0) {
        v[0] = -v[0];
        v[1] = -v[1];
        v[2] = -v[2];
    }
    o[0] = (float)v[0];
    o[1] = (float)v[1];
    o[2] = (float)v[2];
    if (cv) *cv = trace != 0.0 ? (float)fabs(lambda / trace) : 0.0f;
}

__device__ __forceinline__ void acc_sum(NormalAcc &a, const float4 &q) {
    a.sx += (double)q.x;
    a.sy += (double)q.y;
    a.sz += (double)q.z;
    ++a.m;
}
__device__ __forceinline__ void acc_mean(NormalAcc &a) {
    a.mx = a.sx / (double)a.m;
    a.my = a.sy / (double)a.m;
    a.mz = a.sz / (double)a.m;
}
__device__ __forceinline__ void acc_cov(NormalAcc &a, const float4 &q) {
    const double dx = (double)q.x - a.mx, dy = (double)q.y - a.my, dz = (double)q.z - a.mz;
    a.c00 += dx * dx;
    a.c01 += dx * dy;
    a.c02 += dx * dz;
    a.c11 += dy * dy;
    a.c12 += dy * dz;
    a.c22 += dz * dz;
}

struct NormalOut {
    char *normals;            // 3 floats per point at byte stride
    size_t normals_stride;
    char *curvature;          // 1 float per point at byte stride, may be null
    size_t curvature_stride;
    float vx, vy, vz;         // viewpoint
};

// k-search: the KCAP best (d2, index) are kept sorted in registers, the first k of them are used;
// the searched block of cells grows until the k-th distance is safe (as second_nn_kernel)
template <int KCAP>
__global__ __launch_bounds__(128) void knn_normals_kernel(const float4 *__restrict__ pts,
                                                          const int *__restrict__ cell_start,
                                                          const int *__restrict__ pos_of,
                                                          const DevState *__restrict__ ds, int n, int k,
                                                          NormalOut out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GridDesc g = ds->grid;
    const int s = g.ncells > 0 ? pos_of[i] : -1;
    NormalAcc acc = {};
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) {
        p = pts[s];
        const int cx = cell_coord(p.x, g.mn[0], g.h, g.dims[0]);
        const int cy = cell_coord(p.y, g.mn[1], g.h, g.dims[1]);
        const int cz = cell_coord(p.z, g.mn[2], g.h, g.dims[2]);
        const int maxring = max(g.dims[0], max(g.dims[1], g.dims[2]));
        float bd[KCAP];
        int bi[KCAP], bt[KCAP];
        for (int ring = 1; ring <= maxring; ++ring) {
            const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.dims[0] - 1);
            const int y0 = max(cy - ring, 0), y1 = min(cy + ring, g.dims[1] - 1);
            const int z0 = max(cz - ring, 0), z1 = min(cz + ring, g.dims[2] - 1);
#pragma unroll
            for (int u = 0; u < KCAP; ++u) {
                bd[u] = INFINITY;
                bi[u] = 0x7fffffff;
                bt[u] = 0;
            }
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y) {
                    const int row = (z * g.dims[1] + y) * g.dims[0];
                    const int t0 = cell_start[row + x0], t1 = cell_start[row + x1 + 1];
                    for (int t = t0; t < t1; ++t) {
                        const float4 q = pts[t];
                        const float d2 = dist2(p.x, p.y, p.z, q);
                        const int j = __float_as_int(q.w);
                        if (d2 < bd[KCAP - 1] || (d2 == bd[KCAP - 1] && j < bi[KCAP - 1])) {
                            bool lt[KCAP];
#pragma unroll
                            for (int u = 0; u < KCAP; ++u) lt[u] = d2 < bd[u] || (d2 == bd[u] && j < bi[u]);
#pragma unroll
                            for (int u = KCAP - 1; u >= 0; --u) {
                                const bool here = lt[u] && (u == 0 || !lt[u - 1]);
                                const float pd = u > 0 ? bd[u - 1] : 0.f;
                                const int pi = u > 0 ? bi[u - 1] : 0, pt = u > 0 ? bt[u - 1] : 0;
                                bd[u] = !lt[u] ? bd[u] : (here ? d2 : pd);
                                bi[u] = !lt[u] ? bi[u] : (here ? j : pi);
                                bt[u] = !lt[u] ? bt[u] : (here ? t : pt);
                            }
                        }
                    }
                }
            float dk = INFINITY;
#pragma unroll
            for (int u = 0; u < KCAP; ++u) dk = (u == k - 1) ? bd[u] : dk;
            // every point outside the block is at least ring*h away (0.999: slack for the float cell edges)
            if (isfinite(dk) && (double)sqrtf(dk) <= (double)ring * (double)g.h * 0.999) break;
            if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == g.dims[0] - 1 && y1 == g.dims[1] - 1 && z1 == g.dims[2] - 1) break;
        }
#pragma unroll
        for (int u = 0; u < KCAP; ++u)
            if (u < k && isfinite(bd[u])) acc_sum(acc, pts[bt[u]]);
        if (acc.m >= 3) {
            acc_mean(acc);
#pragma unroll
            for (int u = 0; u < KCAP; ++u)
                if (u < k && isfinite(bd[u])) acc_cov(acc, pts[bt[u]]);
        }
    }
    store_normal(acc, p, out.vx, out.vy, out.vz, out.normals, out.normals_stride, out.curvature,
                 out.curvature_stride, i);
}

// radius search on the grid whose cell edge is the radius: neighbors in canonical order
__global__ __launch_bounds__(128) void radius_normals_kernel(const float4 *__restrict__ pts,
                                                             const int *__restrict__ cell_start,
                                                             const int *__restrict__ pos_of,
                                                             const DevState *__restrict__ ds, int n,
                                                             float r2, float rr, NormalOut out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GridDesc g = ds->grid;
    const int s = g.ncells > 0 ? pos_of[i] : -1;
    NormalAcc acc = {};
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) {
        p = pts[s];
        const CellBox b = make_box(g, p.x, p.y, p.z, rr);
        for (int pass = 0; pass < 2; ++pass) {
            for (int z = b.lo[2]; z <= b.hi[2]; ++z)
                for (int y = b.lo[1]; y <= b.hi[1]; ++y) {
                    const int row = (z * g.dims[1] + y) * g.dims[0];
                    const int t0 = cell_start[row + b.lo[0]], t1 = cell_start[row + b.hi[0] + 1];
                    for (int t = t0; t < t1; ++t) {
                        const float4 q = pts[t];
                        if (dist2(p.x, p.y, p.z, q) < r2) {
                            if (pass == 0) acc_sum(acc, q);
                            else acc_cov(acc, q);
                        }
                    }
                }
            if (acc.m < 3) break;
            if (pass == 0) acc_mean(acc);
        }
    }
    store_normal(acc, p, out.vx, out.vy, out.vz, out.normals, out.normals_stride, out.curvature,
                 out.curvature_stride, i);
}

// Ordered compaction of the keypoint flags; also leaves flags[] and the candidate counter clean for the next call.
// ONE launch: scan of the keypoint flags and ordered scatter together, the running total handed from
// block to block through a word per block (single-pass scan with decoupled look-back).  Block x of a view owns the flags
// [4096 x, 4096 x + 4096): it counts them, publishes the count (status AGGREGATE), one wave adds up the words of the blocks
// before it -- 64 at a time, back to the nearest block that already knows its inclusive prefix -- publishes its own inclusive
// prefix (status PREFIX) and scatters.  A word is (call tag << 32 | status << 30 | value): words of earlier calls are simply
// "not there yet", nothing is cleared between calls; the tag is a counter in the view's DevState.  Workgroups are dispatched in index order, so the blocks a block waits
// for are running or done.  Replaces scan_sums + scan_top + scan_apply + compact_kernel: the chain NMS -> keypoint list is 2
// launches instead of 5, which matters most where every launch queues behind another batch's feature kernel.
constexpr unsigned kScanAggregate = 1u, kScanPrefix = 2u;

__device__ __forceinline__ void scan_publish(unsigned long long *word, unsigned epoch, unsigned status, unsigned value) {
    __hip_atomic_store(word, ((unsigned long long)epoch << 32) | ((unsigned long long)status << 30) | (unsigned long long)value,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void compact_scan_kernel(Batch b) {
    constexpr int kChunk = BLOCK * kScanPerThread;            // flags per block
    const ViewDev &v = b.view[blockIdx.y];
    const int poll_limit = v.nd.scan_poll_limit;       // (of the view's handle: kpl_debug_set_scan_poll_limit)
    const int n = v.n, kp_cap = v.kp_cap;
    const int nb = n > 0 ? (n + kChunk - 1) / kChunk : 1;      // blocks of this view (block 0 exists for an empty view too)
    if ((int)blockIdx.x >= nb) return;
    int *flags = v.flags;
    int *skip = v.nd.draws_remove ? v.skip : nullptr;
    unsigned long long *state = v.scan_state;
    // the tag of this call lives ON THE DEVICE (DevState, advanced by the last block below): a call that is replayed from a
    // captured hipGraph gets a fresh tag every time, which a tag passed as a kernel argument would not.  Read ONCE, as an
    // atomic load: the last block rewrites the field while earlier blocks may still be in their look-back
    const unsigned epoch = __hip_atomic_load(&v.ds->scan_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int base = blockIdx.x * kChunk + threadIdx.x * kScanPerThread;
    int f[kScanPerThread];
    int s = 0;
    const bool whole = base + kScanPerThread <= n;
    if (whole) {
        int4 *q = reinterpret_cast<int4 *>(flags + base);
#pragma unroll
        for (int k = 0; k < kScanPerThread / 4; ++k) {
            const int4 x = q[k];
            f[4 * k] = x.x;
            f[4 * k + 1] = x.y;
            f[4 * k + 2] = x.z;
            f[4 * k + 3] = x.w;
            q[k] = make_int4(0, 0, 0, 0);                    // flags[] is left clean for the next call
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) {
            f[k] = base + k < n ? flags[base + k] : 0;
            if (base + k < n) flags[base + k] = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) s += f[k] != 0 ? 1 : 0;
    int total;
    const int excl = block_exclusive_scan<BLOCK>(s, &total);
    __shared__ int carry;
    if (threadIdx.x < kWave) {                               // the first wave: publish, look back, publish
        const int lane = threadIdx.x;
        if (lane == 0) scan_publish(&state[blockIdx.x], epoch, blockIdx.x == 0 ? kScanPrefix : kScanAggregate, (unsigned)total);
        int before = 0;
        bool timed_out = poll_limit < 0 && blockIdx.x != 0;  // (poll_limit < 0: the failure path forced by a test)
        for (int top = (int)blockIdx.x - 1; top >= 0 && !timed_out;) {     // blocks top, top - 1, ... top - 63
            const int j = top - lane;
            unsigned long long w = 0ull;
            bool ready;
            int polls = 0;
            do {                                             // every lane of the window up to the nearest PREFIX must be there
                if (j >= 0) w = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = j < 0 || ((unsigned)(w >> 32) == epoch && ((unsigned)w >> 30) != 0u);
                const unsigned long long have = __ballot(ready);
                const unsigned long long pref = __ballot(ready && j >= 0 && ((unsigned)w >> 30) == kScanPrefix);
                // usable when the lanes below the first PREFIX lane (or all 64) are ready
                const int first = pref ? __builtin_ctzll(pref) : 63;
                const unsigned long long need = first == 63 ? ~0ull : ((2ull << first) - 1ull);
                if ((have & need) == need) {
                    const bool take = j >= 0 && lane <= first;
                    int val = take ? (int)((unsigned)w & 0x3fffffffu) : 0;
                    for (int off = kWave / 2; off > 0; off >>= 1) val += __shfl_xor(val, off);
                    before += val;
                    top = pref ? -1 : top - kWave;
                    break;
                }
                if (++polls > poll_limit) {                  // (never seen.  Liveness rests on the dispatch order of the
                    timed_out = true;                        // workgroups; if that ever fails the call FAILS, it does not guess)
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            } while (true);
        }
        if (timed_out) {
            // the keypoint list of this view is not written from here on; the host sees the failure through
            // DevState::scan_fail (kpl_sync_status -> KPL_ERR_INTERNAL) and, when the last block gets to see it, kp_count = -1.
            // The word published for the blocks behind is well formed (a PREFIX of what is known: they must not wait forever)
            before = 0;
            if (lane == 0) __hip_atomic_store(&v.ds->scan_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) {
            if (blockIdx.x != 0) scan_publish(&state[blockIdx.x], epoch, kScanPrefix, (unsigned)(before + total) & 0x3fffffffu);
            carry = timed_out ? -1 : before;
            if ((int)blockIdx.x == nb - 1) {                 // the last block knows the keypoint count
                const bool failed = v.ds->status != 0 || timed_out ||
                                    __hip_atomic_load(&v.ds->scan_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                *v.kp_count = failed ? -1 : before + total;     // -1: see kpl_sync_status
                *v.cand.count = 0;
                // sorted mode, large neighborhoods, and the word lists of the two-pass walk (32 cursors, each over a 32nd
                // of the array: the fullest one counts): what the call needed, re-armed for the next call
                unsigned long long wmax = 0ull;
                for (int k = 0; k < kWordShards; ++k) {
                    wmax = v.ds->word_cursor[k] > wmax ? v.ds->word_cursor[k] : wmax;
                    v.ds->word_cursor[k] = 0ull;
                }
                v.ds->keys_needed = v.ds->key_cursor + wmax * kWordShards;
                v.ds->words_needed = wmax * kWordShards;
                v.ds->key_cursor = 0ull;
                v.ds->large_seen = v.ds->large_count;
                v.ds->large_count = 0;
                v.ds->huge_count = 0;
                // every block of this launch has published, hence started, hence read the tag: the next call's may be set
                __hip_atomic_store(&v.ds->scan_epoch, epoch + 1u == 0u ? 1u : epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    const bool scatter = carry >= 0;                         // (a block whose look-back failed writes nothing)
    int run = carry + excl;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        const int i = base + k;
        if (skip && i < n) skip[i] = 0;
        if (f[k] != 0) {
            if (scatter && run >= 0 && run < kp_cap) {
                v.kp_idx[run] = i;
                if (v.kp_score) v.kp_score[run] = v.scores[i];     // (scores is set whenever kp_score is)
            }
            ++run;
        }
    }
}

// list of the points whose flag equals `match`, ascending (prefix = scan of that predicate)
__global__ __launch_bounds__(256) void list_match_kernel(Batch b, int match) {
    const ViewDev &v = b.view[blockIdx.y];
    if (!v.nd.draws_remove) return;
    const int *flags = v.flags, *prefix = v.prefix;
    const int n = v.n;
    int *list = v.draw_list, *count = v.draw_count;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *count = prefix[n];
    if (i <= kDrawRounds + 1) v.ds->draws_left[i] = i == 0 ? prefix[n] : 0; // the draws pass of this call: [0] listed, [1] left by draws_adj_kernel, [r + 2] by round r
    if (i < n && flags[i] == match) {
        list[prefix[i]] = i;
        list[n + prefix[i]] = 0;                                              // adjacency count (draws_adj_kernel)
        v.skip[i] = 3;                                                        // kDrawUndecided
    }
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// =============================================================================================
// launch wrappers
// =============================================================================================
// Loads the code objects of libkpl's two kernel files (the runtime loads a code object when the first of its kernels is looked
// up): kpl_create calls this so that no compute() pays for it.  Errors are not fatal here -- a launch would report them.
void preload_organized_normals_code();
void preload_code() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&bbox_kernel));
    preload_organized_normals_code();
    (void)hipGetLastError();
}

// clears `bytes` (a multiple of 4) at p, as a kernel of this library on the caller's stream (DevBuf::ensure)
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
void launch_zero(void *p, size_t bytes, hipStream_t st) {
    const size_t words = bytes / 4;
    if (words == 0) return;
    size_t blocks = (words + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    zero_words_kernel<<<dim3((unsigned)blocks), 256, 0, st>>>(static_cast<uint32_t *>(p), words);
}

void init_dev_state(DevState *host_copy) {
    memset(host_copy, 0, sizeof(*host_copy));
    host_copy->scan_epoch = 1u;         // 0 = "never written" in the words of the compaction's scan
    for (int k = 0; k < 3; ++k) {
        host_copy->bbox[k] = 0xffffffffu;
        host_copy->bbox[3 + k] = 0u;
    }
}

static int max_n(const Batch &b) {
    int m = 0;
    for (int v = 0; v < b.nviews; ++v) m = b.view[v].n > m ? b.view[v].n : m;
    return m;
}

static void run_scan(const ScanJobs &jobs, int nviews, hipStream_t st) {
    int len = 0;
    for (int v = 0; v < nviews; ++v) len = jobs.job[v].len > len ? jobs.job[v].len : len;
    const int nb = len > 0 ? div_up(len, kScanChunk) : 1;
    scan_sums_kernel<<<dim3(nb, nviews), kScanBlock, 0, st>>>(jobs);
    scan_top_kernel<<<dim3(1, nviews), kScanBlock, 0, st>>>(jobs);
    scan_apply_kernel<<<dim3(nb, nviews), kScanBlock, 0, st>>>(jobs);
}

// Index build ("initCompute") of every view of the batch: bounding box -> grid descriptor (on the
// device) -> two-level stable counting sort.  7 launches whatever the batch size (8 with pos_of[]).
size_t btable_ints(int n) { return (size_t)kBins * (size_t)sort_chunks(n > 0 ? n : 1, kSortChunkSmall) + 2 * kBins; }     // (room for the small chunks)

// points per chunk of the index sort's first level for this launch: both halves of the index build of a batch come here
static int sort_chunk_points(const Batch &b) {
    long long total = 0;
    for (int v = 0; v < b.nviews; ++v) total += b.view[v].n;
    return total <= 256 * 1024 ? kSortChunkSmall : kSortChunk;
}

void launch_index(const Batch &b, hipStream_t st) {
    launch_index_points(b, st);
    launch_index_records(b, st);
}

// first half: the kernels that read the POINTS only (bounding box, grid, bucket counts and offsets)
void launch_index_points(const Batch &b, hipStream_t st) {
    const int nv = b.nviews, n = max_n(b);
    if (nv <= 0) return;
    if (n > 0) {
        int blocks = div_up(n, 256);
        if (blocks > 64) blocks = 64;                   // 6 same-address atomics per block: keep them few
        bbox_kernel<<<dim3(blocks, nv), 256, 0, st>>>(b);
    }
    grid_setup_kernel<<<dim3(1, nv), 64, 0, st>>>(b);
    const int chunk_pts = sort_chunk_points(b);
    if (n > 0) bucket_hist_kernel<<<dim3(sort_chunks(n, chunk_pts), nv), kWave, 0, st>>>(b, chunk_pts);
    if (nv >= 2 && n >= 128 * 1024) {   // a batch of LARGE views: one wave per workgroup (see kScanBlock): bench 1 969 -> 1 980 Mpoints/s; for
                                        // batches of small views the 1 025 one-wave blocks per view cost more than they free: 64 views
                                        // of 63 k points 1 490 against 1 553-1 602 (profiles/r06_notes.md)
        bucket_total_kernel<<<dim3(kBins, nv), kWave, 0, st>>>(b, chunk_pts);
        bucket_offsets_kernel<kWave><<<dim3(kBins, nv), kWave, 0, st>>>(b, chunk_pts);
    } else {                // one view alone on the GPU: the shortest chain (0.046 against 0.049 ms of index build on 62 k points)
        bucket_total_kernel<<<dim3(div_up(kBins, 256 / kWave), nv), 256, 0, st>>>(b, chunk_pts);
        bucket_offsets_kernel<256><<<dim3(div_up(kBins, 256 / kWave), nv), 256, 0, st>>>(b, chunk_pts);
    }
}

// second half: from the scatter on the NORMALS are read too (the 32-byte record travels with the key)
void launch_index_records(const Batch &b, hipStream_t st) {
    const int nv = b.nviews, n = max_n(b);
    if (nv <= 0) return;
    const int chunk_pts = sort_chunk_points(b);
    if (n > 0) bucket_scatter_kernel<<<dim3(sort_chunks(n, chunk_pts), nv), kWave, 0, st>>>(b, chunk_pts);
    // cells of a bucket held in LDS at a time: what the largest cell table of the batch can need, at most
    // 1024 (two cells per thread in the workgroup's scan; kSortWaves counters per cell)
    int cap = 0;
    for (int v = 0; v < nv; ++v) cap = b.view[v].cells_cap > cap ? b.view[v].cells_cap : cap;
    int lds_cells = 64, nbits = 6;
    while (lds_cells < 1024 && (long long)lds_cells * kBuckets < (long long)cap) {
        lds_cells <<= 1;
        ++nbits;
    }
    cell_sort_store_kernel<<<dim3(kBuckets, nv), kSortWaves * kWave,
                             sizeof(int) * (size_t)kSortWaves * (size_t)(lds_cells + kTagSlots), st>>>(b, lds_cells, nbits);
    bool want = false;
    for (int v = 0; v < nv; ++v) want |= b.view[v].want_pos_of != 0;
    if (want && n > 0) pos_of_kernel<<<dim3(div_up(n, 256), nv), 256, 0, st>>>(b);
}

void launch_pos_of(const Batch &b, hipStream_t st) {
    const int n = max_n(b);
    if (b.nviews > 0 && n > 0) pos_of_kernel<<<dim3(div_up(n, 256), b.nviews), 256, 0, st>>>(b);
}

size_t feat_bytes(int n, int F) { return sizeof(float) * (size_t)div_up(n > 0 ? n : 1, kLanes) * kLanes * (size_t)(F > 0 ? F : 1); }

size_t pts_bytes(int n) { return sizeof(float4) * ((size_t)(n > 0 ? n : 1) + kStepW); }

static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

// Accept words per point (what a point may collect between two drains; a neighborhood that needs more simply
// searches and drains in several rounds): 24 = 768 candidates.  Measured 12 / 16 / 24 / 32 on every config
// (profiles/r02_notes.md): 24 is best or level everywhere -- the larger the neighborhoods the more (500 k
// points at r = 10 mr: 0.639 / 0.593 / 0.499 ms; config 5: 1.29 / 1.31 / 1.22; one 62 k-point view: 0.086 /
// 0.076 / 0.077; 8 views of 200 k: 0.654 / - / 0.653 / 0.770) -- with ONE exception that is not taken: 12 words
// together with a register cap of 88 (five waves per SIMD instead of four) run the 8-view batch in 0.621 ms,
// and every launch with larger neighborhoods 10-25 % slower; the host cannot know the neighborhood size.
// Fewer words for the largest histograms so that a handful of waves still fit a CU (160 KB of LDS).
constexpr int kLdsPerCu = 160 * 1024;
// Round 5: the host now DOES know the neighborhood size (FeatDesc::words, chosen by api.cpp from the mean K_f the handle's
// last call measured), and the kernel has come down to 96 registers since, so that fewer words are more resident waves (24
// words: 10 KB per wave = 4 waves per SIMD; 16 and fewer: 5) -- and a smaller list also makes the waves of a launch alternate
// between search (texture path) and drain (VALU) at different times instead of all searching first.  Feature kernel alone
// on the GPU, 8 x 200 k points at K_f = 69: 24 words 0.625 ms, 16: 0.588, 14: 0.566, 12: 0.563; the bench 1 900 -> 1 950
// Mpoints/s.  One 500 k-point view, 24 / 16 / 12 words: K_f = 32: 0.146 / 0.140 / 0.135 ms, 69: 0.217 / 0.208 / 0.200, 124:
// 0.325 / 0.310 / 0.340, 192: 0.464 / 0.525 / 0.542; cheff000 (K_f = 100, up to 159): 12 words cost 7 %, 16 nothing.
template <int G>
static int accept_words(int F, int wanted = 0) {
    int e = wanted > 0 && wanted < 24 ? wanted : 24;
    while (e > 4 && feature_lds_bytes<G>(F, e) * 6 > (size_t)kLdsPerCu) e -= 4;
    return e;
}

// Keys per point and pass of the sorted-search mode = what the register sort holds (128: 16 KB per wave of 16
// points; longer neighborhoods take several passes)
static int sorted_list_keys(int) { return kSortedListKeys; }

// Geometry of the forest kernel: one workgroup per CU; LDS = the top of the forest (at most
// kForestNodeBytes) + one F x 64 float slice per wave, as many waves as then fit (2..16).
constexpr size_t kForestLds = 156 * 1024, kForestNodeBytes = 64 * 1024;
struct ForestLaunch {
    int waves, nlds_cap;
    size_t lds;
};
static ForestLaunch forest_launch(int maxF, int max_nodes) {
    ForestLaunch fl;
    const size_t slice = sizeof(float) * (size_t)maxF * kLanes;
    size_t node_bytes = sizeof(uint2) * (size_t)(max_nodes > 0 ? max_nodes : 1);
    if (node_bytes > kForestNodeBytes) node_bytes = kForestNodeBytes;
    long long w = ((long long)kForestLds - (long long)node_bytes) / (long long)slice;
    if (w < 2) {                                   // huge histograms: two waves, the nodes get what is left
        w = 2;
        node_bytes = kForestLds - 2 * slice;       // F <= 255: 2 slices are at most 130 KB
    }
    if (w > 16) w = 16;
    fl.waves = (int)w;
    fl.nlds_cap = (int)(node_bytes / sizeof(uint2));
    fl.lds = sizeof(uint2) * (size_t)fl.nlds_cap + slice * (size_t)fl.waves;
    return fl;
}


// scoring ("runForest") of every view of the batch, first kernel: histogram features -> feat
void launch_feature_stage(const Batch &b, hipStream_t st) {
    const int n = max_n(b);
    if (b.nviews <= 0 || n <= 0) return;
    int maxF = 1;
    bool stats = false;
    for (int v = 0; v < b.nviews; ++v) {
        maxF = b.view[v].f.F > maxF ? b.view[v].f.F : maxF;
        stats |= b.view[v].stats != nullptr;
    }
    // the views of the batch by how their features are computed: every kernel of the stage skips the views of the others
    bool sorted = false, lanes2 = false, lanes4 = false, two2 = false, two4 = false;
    for (int v = 0; v < b.nviews; ++v) {
        const FeatDesc &f = b.view[v].f;
        if (f.sorted) sorted = true;                 // (walk == 1 there: the sorted-words kernels, launched in the sorted branch)
        else if (f.walk == kWalkTwoPass) (f.lanes == 4 ? two4 : two2) = true;
        else (f.lanes == 4 ? lanes4 : lanes2) = true;
    }
    // accept words per point: what the views of that walk ask for (the largest of them: a launch has one list capacity)
    // -- and only for a launch that has more waves than are resident with 24 words (4 per SIMD): a 63 k-point view alone is two
    // waves per SIMD whatever the lists, and short lists only make it drain more often (0.063 -> 0.071 ms at K_f = 70)
    auto words_wanted = [&](int lanes) {
        int e = 0;
        long long waves = 0;
        for (int v = 0; v < b.nviews; ++v) {
            const FeatDesc &f = b.view[v].f;
            if (f.sorted || f.walk != kWalkLanes || f.lanes != lanes) continue;
            const int w = f.words > 0 ? f.words : 24;
            e = w > e ? w : e;
            waves += div_up(b.view[v].n, kLanes / lanes);
        }
        return waves > 4ll * 4 * cu_count() ? e : 24;
    };
    if (lanes2) {
        const int ecap = accept_words<2>(maxF, words_wanted(2));
        const size_t lds = feature_lds_bytes<2>(maxF, ecap);
        const dim3 grid(div_up(n, kLanes) * 2, b.nviews);
        if (stats) feature_kernel<true, 2><<<grid, kLanes, lds, st>>>(b, maxF, ecap);
        else feature_kernel<false, 2><<<grid, kLanes, lds, st>>>(b, maxF, ecap);
    }
    if (lanes4) {
        const int ecap = accept_words<4>(maxF, words_wanted(4));
        const size_t lds = feature_lds_bytes<4>(maxF, ecap);
        const dim3 grid(div_up(n, kLanes) * 4, b.nviews);
        if (stats) feature_kernel<true, 4><<<grid, kLanes, lds, st>>>(b, maxF, ecap);
        else feature_kernel<false, 4><<<grid, kLanes, lds, st>>>(b, maxF, ecap);
    }
    // two passes (large neighborhoods): the search for every such view (eight lanes per point: the lists of a wave's 8
    // points are interleaved), then the drain with the lanes per point of the view
    if (two2 || two4) {
        const dim3 sgrid(div_up(n, kLanes) * kSearchGroup, b.nviews);
        if (stats) feature_search_kernel<true><<<sgrid, kLanes, 0, st>>>(b, 0);
        else feature_search_kernel<false><<<sgrid, kLanes, 0, st>>>(b, 0);
        const int stride = kLanes / kSearchGroup;
        if (two2)
            feature_drain_kernel<2><<<dim3(div_up(n, kLanes) * 2, b.nviews), kLanes, sizeof(float) * (size_t)maxF * (kLanes / 2), st>>>(b, maxF, stride);
        if (two4)
            feature_drain_kernel<4><<<dim3(div_up(n, kLanes) * 4, b.nviews), kLanes, sizeof(float) * (size_t)maxF * (kLanes / 4), st>>>(b, maxF, stride);
    }
    if (sorted) {       // the views in sorted-search mode (each kernel skips the views of the other mode)
        // the register-sort kernel lists the points with large neighborhoods instead of scoring them, the collect / add pair
        // takes them (both return at once where there are none)
        // keys per point of the lists in LDS: what the views' handles ask for (FeatDesc::lcap, from the longest neighborhood
        // their earlier calls saw; the register sort holds 128 either way) -- 96 keys instead of 128 are 10 instead of 8 waves
        // per CU (8 x 200 k points at 6 mr: 2.11 -> 1.84 ms, profiles/r04_sorted_lcap.jsonl)
        int lcap = 8;
        for (int v = 0; v < b.nviews; ++v)
            if (b.view[v].f.sorted) lcap = std::max(lcap, b.view[v].f.lcap > 0 && b.view[v].f.lcap <= kSortedListKeys ? b.view[v].f.lcap : kSortedListKeys);
        const size_t lds = sorted_view_lds_bytes<kSortGroup>(maxF, kSortWords, lcap);
        const dim3 grid(div_up(n, kLanes) * kSortGroup, b.nviews);
        // views of about the same size are dealt to the XCDs (view_block): what a view reads at random then stays in one L2
        int min_n = n;
        for (int k = 0; k < b.nviews; ++k) min_n = b.view[k].n < min_n ? b.view[k].n : min_n;
        const int by_xcd = (b.nviews >= 2 && (long long)min_n * 10 >= (long long)n * 9) ? 1 : 0;
        bool reg_views = false, word_views = false;     // which of the two front kernels the sorted views of the batch take
        for (int v = 0; v < b.nviews; ++v)
            if (b.view[v].f.sorted) (b.view[v].f.walk == kWalkTwoPass ? word_views : reg_views) = true;
        if (reg_views) {
            if (stats) feature_sorted_kernel<true><<<grid, kLanes, lds, st>>>(b, maxF, kSortWords, lcap, by_xcd);
            else feature_sorted_kernel<false><<<grid, kLanes, lds, st>>>(b, maxF, kSortWords, lcap, by_xcd);
        }
        if (word_views) {       // 125 .. ~250 neighbors per point: the search pass of the two-pass walk, then eight lanes per point
            const dim3 sgrid(div_up(n, kLanes) * kSearchGroup, b.nviews);
            if (stats) feature_search_kernel<true><<<sgrid, kLanes, 0, st>>>(b, 1);
            else feature_search_kernel<false><<<sgrid, kLanes, 0, st>>>(b, 1);
            // (FeatDesc::lcap of such a view: 256 positions per point, or 512 -- twice the LDS, half the waves per CU)
            bool w256 = false, w512 = false;
            for (int v = 0; v < b.nviews; ++v)
                if (b.view[v].f.sorted && b.view[v].f.walk == kWalkTwoPass) (b.view[v].f.lcap > kWordsKeys ? w512 : w256) = true;
            const dim3 wgrid(div_up(n, kLanes) * kWordsGroup, b.nviews);
            if (w256) {
                const size_t wlds = (sizeof(float) * (size_t)maxF + sizeof(unsigned) * (size_t)kWordsKeys) * (size_t)(kLanes / kWordsGroup);
                if (stats) sorted_words_kernel<true, 32><<<wgrid, kLanes, wlds, st>>>(b, maxF, by_xcd);
                else sorted_words_kernel<false, 32><<<wgrid, kLanes, wlds, st>>>(b, maxF, by_xcd);
            }
            if (w512) {
                const size_t wlds = (sizeof(float) * (size_t)maxF + sizeof(unsigned) * (size_t)(2 * kWordsKeys)) * (size_t)(kLanes / kWordsGroup);
                if (stats) sorted_words_kernel<true, 64><<<wgrid, kLanes, wlds, st>>>(b, maxF, by_xcd);
                else sorted_words_kernel<false, 64><<<wgrid, kLanes, wlds, st>>>(b, maxF, by_xcd);
            }
        }
        // persistent: as many workgroups of four waves as are resident at once (every wave takes the same share of the list:
        // workgroups that start when others have finished would double the kernel's time)
        static int wave_wgs_per_cu = 0;         // (every thread that gets here computes the same value)
        if (wave_wgs_per_cu == 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sorted_collect_wave_kernel, kWaveCollectWaves * kWave, 0) != hipSuccess || nb < 1) nb = 4;
            wave_wgs_per_cu = nb;
        }
        int wwgs = div_up(cu_count() * wave_wgs_per_cu, b.nviews);
        if (wwgs > div_up(n, kWaveCollectWaves)) wwgs = div_up(n, kWaveCollectWaves);
        bool all_huge = true;               // every sorted view lists all of its points for the workgroup kernel: nothing for this one
        for (int v = 0; v < b.nviews; ++v) all_huge = all_huge && (!b.view[v].f.sorted || b.view[v].f.all_large == 2);
        if (!all_huge) sorted_collect_wave_kernel<<<dim3(wwgs, b.nviews), kWaveCollectWaves * kWave, 0, st>>>(b, by_xcd);
        int wgs = div_up(cu_count() * 4, b.nviews);             // persistent: four workgroups per CU (37 KB of LDS each)
        if (wgs > n) wgs = n;
        sorted_collect_kernel<<<dim3(wgs, b.nviews), kCollectThreads, 0, st>>>(b);
        int awgs = div_up(n, kLanes / 4);                         // a wave per 16 points at most, persistent beyond 16 per CU
        if (awgs > cu_count() * 16) awgs = cu_count() * 16;
        const dim3 agrid(awgs, b.nviews);
        const size_t alds = sizeof(float) * (size_t)maxF * (kLanes / 2);
        if (stats) sorted_add_kernel<true><<<agrid, kLanes, alds, st>>>(b, maxF, by_xcd);
        else sorted_add_kernel<false><<<agrid, kLanes, alds, st>>>(b, maxF, by_xcd);
    }
}

// second kernel: feat -> forest response (score_sorted, scores) and the NMS candidates
void launch_forest_stage(const Batch &b, hipStream_t st) {
    const int n = max_n(b);
    if (b.nviews <= 0 || n <= 0) return;
    int maxF = 1, max_nodes = 1, min_trees = 1 << 30;
    bool stats = false, any_order = true;
    for (int v = 0; v < b.nviews; ++v) {
        maxF = b.view[v].f.F > maxF ? b.view[v].f.F : maxF;
        max_nodes = b.view[v].forest.ntop > max_nodes ? b.view[v].forest.ntop : max_nodes;      // what is staged in LDS
        min_trees = b.view[v].forest.ntrees < min_trees ? b.view[v].forest.ntrees : min_trees;
        any_order &= b.view[v].forest.order_free != 0;
        stats |= b.view[v].stats != nullptr;
    }
    // several lanes per point (forest_split_kernel) when every forest of the batch is chained (forest.h: its sum is exact in
    // any order, it has trees to share out -- >= 40 --, and a whole wave of its histograms -- F >= 32 -- would crowd the
    // waves out of LDS).  Lanes per point x walks per lane -> forest kernel ms on config 5 (1 M points, 100 trees), with the
    // tree queue of round 2: 4 x 3 1.57, 4 x 2 1.63, 4 x 5 1.78, 8 x 3 1.74, 8 x 5 2.07, 2 x 3 2.08, 2 x 5 2.13, one lane
    // per point with 10 walks 2.15; chained: 4 x 3 1.02, 4 x 4 0.94, 4 x 5 0.97, 4 x 6 0.99 (profiles/r03_notes.md)
    constexpr int G = 4;
    int chain = b.view[0].forest.chain;
    for (int v = 0; v < b.nviews; ++v) chain = b.view[v].forest.chain == chain ? chain : 0;
    const int ways = chain / G;
    if (any_order && ways == kChainWays && chain == ways * G) {
        const size_t slice = sizeof(float) * (size_t)maxF * (kLanes / G);
        size_t node_bytes = sizeof(uint2) * (size_t)max_nodes;
        if (node_bytes > kForestNodeBytes) node_bytes = kForestNodeBytes;
        long long w = ((long long)kForestLds - (long long)node_bytes) / (long long)slice;
        if (w > 16) w = 16;
        const int nlds_cap = (int)(node_bytes / sizeof(uint2));
        const size_t lds = node_bytes + slice * (size_t)w;
        int wgs = div_up(cu_count(), b.nviews);
        const int wgs_max = div_up(div_up(n, kLanes / G), (int)w);
        if (wgs > wgs_max) wgs = wgs_max;
        const dim3 sgrid(wgs, b.nviews);
        if (stats) forest_split_kernel<true><<<sgrid, (int)w * kLanes, lds, st>>>(b, maxF, nlds_cap, G);
        else forest_split_kernel<false><<<sgrid, (int)w * kLanes, lds, st>>>(b, maxF, nlds_cap, G);
        return;
    }
    // small order-free forests that fit the LDS whole: two lanes per point (forest_pair_kernel)
    {
        int max_all = 1, max_trees = 0;
        for (int v = 0; v < b.nviews; ++v) {
            max_all = b.view[v].forest.nnodes > max_all ? b.view[v].forest.nnodes : max_all;
            max_trees = b.view[v].forest.ntrees > max_trees ? b.view[v].forest.ntrees : max_trees;
        }
        constexpr int G = 2;       // (4 lanes per point, 3 trees each: 0.172 instead of 0.133 ms, profiles/r03_notes.md)
        const size_t node_bytes = sizeof(uint2) * (size_t)max_all, slice = sizeof(float) * (size_t)maxF * (kLanes / G);
        // (for batches: a single view alone on the GPU is 10-25 % faster through the one-lane kernel -- 0.044 against 0.050 ms
        // at 200 k points, 0.10 against 0.12 at 500 k --, a batch of 8 is 13 % faster through this one even alone)
        if (b.nviews >= 2 &&
            any_order && max_trees <= 2 * kPairWays && node_bytes <= kForestNodeBytes && node_bytes + 16 * slice <= kForestLds) {
            const int waves = 16;
            int wgs = div_up(cu_count(), b.nviews);
            const int wgs_max = div_up(div_up(n, kLanes / G), waves);
            if (wgs > wgs_max) wgs = wgs_max;
            const size_t lds = node_bytes + slice * (size_t)waves;
            const dim3 grid(wgs, b.nviews);
            if (max_trees <= 10) {
                if (stats) forest_pair_kernel<true, 2, 5><<<grid, waves * kLanes, lds, st>>>(b, maxF, max_all);
                else forest_pair_kernel<false, 2, 5><<<grid, waves * kLanes, lds, st>>>(b, maxF, max_all);
            } else {
                if (stats) forest_pair_kernel<true, 2, kPairWays><<<grid, waves * kLanes, lds, st>>>(b, maxF, max_all);
                else forest_pair_kernel<false, 2, kPairWays><<<grid, waves * kLanes, lds, st>>>(b, maxF, max_all);
            }
            return;
        }
    }
    const ForestLaunch fl = forest_launch(maxF, max_nodes);
    int wgs = div_up(cu_count(), b.nviews);                   // persistent: about one workgroup per CU
    const int wgs_max = div_up(div_up(n, kLanes), fl.waves);
    if (wgs > wgs_max) wgs = wgs_max;
    const dim3 fgrid(wgs, b.nviews);
    if (stats) forest_kernel<true><<<fgrid, fl.waves * kLanes, fl.lds, st>>>(b, maxF, fl.nlds_cap);
    else forest_kernel<false><<<fgrid, fl.waves * kLanes, fl.lds, st>>>(b, maxF, fl.nlds_cap);
}

void launch_features(const QueryBatch &qb, hipStream_t st) {
    int max_m = 0, maxF = 1;
    bool canonical = false, sorted = false;
    for (int v = 0; v < qb.nviews; ++v) {
        const QueryView &q = qb.view[v];
        if (q.m <= 0) continue;
        max_m = q.m > max_m ? q.m : max_m;
        maxF = q.f.F > maxF ? q.f.F : maxF;
        (q.f.sorted ? sorted : canonical) = true;
    }
    if (max_m <= 0) return;
    if (canonical) {
        const int ecap = accept_words<kGroup>(maxF);
        features_kernel<<<dim3(div_up(max_m, kLanes / kGroup), qb.nviews), kLanes, feature_lds_bytes<kGroup>(maxF, ecap), st>>>(qb, maxF, ecap);
    }
    if (sorted) {
        const int lcap = sorted_list_keys(maxF);
        features_sorted_kernel<<<dim3(div_up(max_m, kLanes / kSortGroup), qb.nviews), kLanes,
                                 sorted_lds_bytes<kSortGroup>(maxF, kSortWords, lcap), st>>>(qb, maxF, kSortWords, lcap);
    }
}

// NMS, draws pass (if any view asks for it), flag scan and ordered compaction of every view
void launch_post(const Batch &b, hipStream_t st) {
    const int nv = b.nviews, n = max_n(b);
    if (nv <= 0) return;
    bool stats = false, draws = false, nms = false;
    for (int v = 0; v < nv; ++v) {
        stats |= b.view[v].stats != nullptr;
        draws |= b.view[v].nd.draws_remove != 0;
        nms |= b.view[v].nd.non_maxima != 0;
    }
    ScanJobs jobs;
    jobs.zero_in = 0;
    if (n > 0 && nms) {
        // lanes per candidate: a batch is scored for throughput (4), a single view for latency (16, 8 when it has many candidates)
        const int L = nv >= 2 ? 4 : 16;
        int blocks = div_up(n, 256 / L);                // at most one group per point ...
        const int most = nv >= 2 ? 128 : 2048;          // ... but a few waves per SIMD are plenty
        if (blocks > most) blocks = most;
        const int many = 5 * blocks * (256 / 16) / 4;   // more candidates than the groups of 16 take in about one round
        if (nv >= 2) {      // a batch: workgroups of ONE wave (see kScanBlock): nms_kernel avg 118 -> 101 us, max 641 -> 186 under the bench
            const dim3 grid(blocks * 4, nv);
            if (stats) nms_kernel<true, 4, 0><<<grid, kWave, 0, st>>>(b, 0);
            else nms_kernel<false, 4, 0><<<grid, kWave, 0, st>>>(b, 0);
        } else {
            const dim3 grid(blocks, nv);
            if (stats) nms_kernel<true, 16, 8><<<grid, 256, 0, st>>>(b, many);
            else nms_kernel<false, 16, 8><<<grid, 256, 0, st>>>(b, many);
        }
        if (draws) {
            jobs.match = 2;
            for (int v = 0; v < nv; ++v) {
                const ViewDev &w = b.view[v];
                jobs.job[v] = ScanJob{w.flags, w.prefix, nullptr, nullptr, w.nd.draws_remove ? w.n : 0, w.scan_tmp};
            }
            run_scan(jobs, nv, st);
            list_match_kernel<<<dim3(div_up(n, 256), nv), 256, 0, st>>>(b, 2);
            int dblocks = div_up(n, 256 / kDrawLanes);
            if (dblocks > 256) dblocks = 256;
            draws_adj_kernel<<<dim3(dblocks, nv), 256, 0, st>>>(b);
            for (int r = 0; r < kDrawRounds; ++r) draws_round_kernel<<<dim3(dblocks, nv), 256, 0, st>>>(b, r);
            int lds_words = div_up(n, 16);                                    // 2 bits of state per listed maximum
            if (lds_words > 20 * 1024) lds_words = 20 * 1024;                 // 80 KB of states (327 k listed maxima) + 8 x 8 KB of rows
            draws_rest_kernel<<<dim3(1, nv), kRestWaves * kWave,
                                sizeof(uint32_t) * (size_t)lds_words + sizeof(int) * (size_t)kRestWaves * kWave * kDrawAdj, st>>>(b, lds_words);
        }
    }
    if (nv >= 2) compact_scan_kernel<kCompactBatchBlock><<<dim3(n > 0 ? div_up(n, kCompactBatchBlock * kScanPerThread) : 1, nv), kCompactBatchBlock, 0, st>>>(b);
    else compact_scan_kernel<kScanBlock><<<dim3(n > 0 ? div_up(n, kScanChunk) : 1, nv), kScanBlock, 0, st>>>(b);
}

size_t scan_state_bytes(int n) { return sizeof(unsigned long long) * ((size_t)(n > 0 ? n : 1) / (kCompactBatchBlock * kScanPerThread) + 2); }      // (the smaller blocks' count)

void launch_resolution(const float4 *pts, const int *cell_start, const int *pos_of, const DevState *ds,
                       int n, float *val, double *out, void *scratch, hipStream_t st) {
    if (n > 0) second_nn_kernel<<<div_up(n, 256), 256, 0, st>>>(pts, cell_start, pos_of, ds, n, val);
    SumPart *part = static_cast<SumPart *>(scratch);
    resolution_partial_kernel<<<kSumBlocks, 256, 0, st>>>(val, n, part);
    resolution_finish_kernel<<<1, 256, 0, st>>>(part, kSumBlocks, out);
    ordered_sum_kernel<<<1, 64, 0, st>>>(val, n, out);
}

size_t resolution_scratch_bytes() { return sizeof(SumPart) * kSumBlocks; }

void launch_normals(const float4 *pts, const int *cell_start, const int *pos_of, const DevState *ds, int n,
                    int k, float r2, float rr, const float *viewpoint, char *normals, size_t normals_stride,
                    char *curvature, size_t curvature_stride, hipStream_t st) {
    if (n <= 0) return;
    NormalOut out{normals, normals_stride, curvature, curvature_stride, viewpoint[0], viewpoint[1], viewpoint[2]};
    const int blocks = div_up(n, 128);
    if (k <= 0) radius_normals_kernel<<<blocks, 128, 0, st>>>(pts, cell_start, pos_of, ds, n, r2, rr, out);
    else if (k <= 16) knn_normals_kernel<16><<<blocks, 128, 0, st>>>(pts, cell_start, pos_of, ds, n, k, out);
    else knn_normals_kernel<32><<<blocks, 128, 0, st>>>(pts, cell_start, pos_of, ds, n, k, out);
}

}  // namespace kpl
