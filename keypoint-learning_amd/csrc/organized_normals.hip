// Normals of an ORGANIZED cloud: pcl::IntegralImageNormalEstimation as the reference's fallback drives
// it (include/impl/KeypointLearning.hpp:138-145: SIMPLE_3D_GRADIENT, setNormalSmoothingSize(5.0)),
// on the device.  PCL 1.8.0 is absent from /root/reference; the algorithm is restated from its
// published source (features/impl/integral_image_normal.hpp, integral_image2D.hpp; the same
// restatement, loop by loop on the CPU, is what the parity tests check these kernels against), and
// the kernels compute the same floats in the same order:
//   change map   every pixel pair (right, lower) on its own thread; all writes are the same zero
//   distance map the two chamfer passes are recurrences along the row (left / right neighbor) and
//                across rows (three neighbors of the previous row): one thread per row, row i two
//                columns behind row i - 1, one workgroup, a barrier per step; the neighbor row's
//                values travel through a 4-deep ring in LDS.  width + 2 * rows steps per pass
//   integral img the same skew for cur[c + 1] = (prev[c + 1] + cur[c]) - prev[c] (+ point): double
//                additions are kept in PCL's order, so no parallel prefix sum
//   normals      one thread per pixel: four rectangle sums each for the two gradients, cross product
//                and normalisation in double, flip towards the viewpoint in float
// Not part of the scoring hot path (kernels.hip); tens of microseconds do not matter here, the order of
// the floating-point operations does.
#include "organized_normals.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace kpl {
namespace {

constexpr int kRows = 1024;        // rows (threads) of one band of the skewed recurrences

__device__ __forceinline__ const float *pixel(const OrganizedView &v, size_t index) {
    return reinterpret_cast<const float *>(v.xyz + index * v.xs);
}

__global__ __launch_bounds__(256) void on_init_kernel(OrganizedView v) {
    const size_t n = (size_t)v.W * (size_t)v.H;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        v.change[i] = 255;
        float *o = reinterpret_cast<float *>(v.normals + i * v.ns);
        o[0] = o[1] = o[2] = NAN;
        if (v.curvature) *reinterpret_cast<float *>(v.curvature + i * v.cs) = NAN;
    }
    if (i < (size_t)3 * (size_t)(v.W + 1)) v.ii[i] = 0.0;          // row 0 of the integral image
}

// integral_image_normal.hpp computeFeature, "compute depth-change map"
__global__ __launch_bounds__(256) void on_change_kernel(OrganizedView v) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
    if (ci >= v.W - 1 || ri >= v.H - 1) return;
    const size_t index = (size_t)ri * (size_t)v.W + (size_t)ci;
    const float depth = pixel(v, index)[2], depthR = pixel(v, index + 1)[2], depthD = pixel(v, index + (size_t)v.W)[2];
    const float limit = ((20.0f * 0.001f) * (fabsf(depth) + 1.0f) * 2.0f);
    if (fabs((double)(depth - depthR)) > (double)limit || !isfinite(depth) || !isfinite(depthR)) {
        v.change[index] = 0;
        v.change[index + 1] = 0;
    }
    if (fabs((double)(depth - depthD)) > (double)limit || !isfinite(depth) || !isfinite(depthD)) {
        v.change[index] = 0;
        v.change[index + (size_t)v.W] = 0;
    }
}

__global__ __launch_bounds__(256) void on_dist_init_kernel(OrganizedView v) {
    const size_t n = (size_t)v.W * (size_t)v.H;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v.dist[i] = v.change[i] == 0 ? 0.0f : (float)(v.W + v.H);
}

// One chamfer pass.  Logical row i = 0 .. H - 2 is image row 1 + i (forward) or H - 2 - i (backward),
// logical column k = 0 .. W - 2 is image column 1 + k resp. W - 2 - k; "behind" is the column handled just
// before in the same row, "ahead" the next one.  PCL's
//   forward   min(min(prev[ci - 1] + 1.4, prev[ci] + 1), min(cur[ci - 1] + 1, prev[ci + 1] + 1.4))
//   backward  min(min(next[ci - 1] + 1.4, next[ci] + 1), min(cur[ci + 1] + 1, next[ci + 1] + 1.4))
// are both the minimum of neighbor[behind] + 1.4, neighbor[here] + 1, neighbor[ahead] + 1.4 and cur[behind] + 1
// (a minimum of floats without NaN does not depend on the order).  Pixels of the edge column (0 forward,
// W - 1 backward) are never written by the pass.  At the last logical column "neighbor[ahead]" is, by PCL's
// loop bounds, one element outside the neighbor row: the current row's own edge pixel (rows are contiguous).
template <bool FORWARD>
__global__ __launch_bounds__(kRows) void on_dist_pass_kernel(OrganizedView v) {
    __shared__ float ring[kRows][4];
    const int W = v.W, H = v.H, nrows = H - 1, ncols = W - 1, j = threadIdx.x;
    const int edge = FORWARD ? 0 : W - 1;
    for (int band = 0; band < nrows; band += kRows) {
        const int i = band + j;
        const bool have = i < nrows;
        const int row = FORWARD ? 1 + i : H - 2 - i, nrow = FORWARD ? row - 1 : row + 1;
        float *cur = v.dist + (size_t)(have ? row : 0) * (size_t)W;
        const float *nb = v.dist + (size_t)(have ? nrow : 0) * (size_t)W;
        const int rows_here = min(kRows, nrows - band);
        float nb_behind = 0.f, nb_here = 0.f, cur_behind = 0.f;
        const float own_edge = have ? cur[edge] : 0.f;
        for (int s = 0; s < ncols + 2 * (rows_here - 1); ++s) {
            const int k = s - 2 * j;
            const bool active = have && k >= 0 && k < ncols;
            float result = 0.f;
            if (active) {
                const int col = FORWARD ? 1 + k : W - 2 - k, ahead = FORWARD ? col + 1 : col - 1;
                if (k == 0) {                      // window at the start of the row
                    nb_behind = nb[edge];          // static during the pass
                    nb_here = j == 0 ? nb[col] : ring[j - 1][0];
                    cur_behind = own_edge;
                }
                float nb_ahead;
                if (k == ncols - 1) nb_ahead = own_edge;
                else nb_ahead = j == 0 ? nb[ahead] : ring[j - 1][(k + 1) & 3];
                const float center = cur[col];
                const float a = fminf(nb_behind + 1.4f, nb_here + 1.0f), b = fminf(cur_behind + 1.0f, nb_ahead + 1.4f);
                const float m = fminf(a, b);
                result = m < center ? m : center;
                if (m < center) cur[col] = m;
                nb_behind = nb_here;
                nb_here = nb_ahead;
                cur_behind = result;
            }
            __syncthreads();                       // every read of the ring of this step is done
            if (active) ring[j][k & 3] = result;
            __syncthreads();
        }
        __threadfence_block();                     // the last row of the band is the next band's neighbor row
        __syncthreads();
    }
}

// integral_image2D.hpp computeIntegralImages (first order): row r of the image -> row r + 1 of ii
__global__ __launch_bounds__(kRows) void on_integral_kernel(OrganizedView v) {
    __shared__ double ring[kRows][2][3];      // out of (row, c) = ii[row + 1][c + 1], kept for the two steps until the next row is at c
    const int W = v.W, H = v.H, j = threadIdx.x;
    const size_t pitch = 3 * (size_t)(W + 1);
    for (int band = 0; band < H; band += kRows) {
        const int r = band + j;
        const bool have = r < H;
        const double *prev = v.ii + (size_t)(have ? r : 0) * pitch;
        double *cur = v.ii + (size_t)(have ? r + 1 : 1) * pitch;
        const int rows_here = min(kRows, H - band);
        double prev_c[3] = {0, 0, 0}, cur_c[3] = {0, 0, 0};         // prev[c], cur[c]: both 0 at c = 0
        if (have) cur[0] = cur[1] = cur[2] = 0.0;
        for (int s = 0; s < W + 2 * (rows_here - 1); ++s) {
            const int c = s - 2 * j;
            const bool active = have && c >= 0 && c < W;
            double out[3] = {0, 0, 0};
            if (active) {
                double prev_n[3];
                for (int a = 0; a < 3; ++a) prev_n[a] = j == 0 ? prev[3 * (size_t)(c + 1) + a] : ring[j - 1][c & 1][a];
                const float *e = pixel(v, (size_t)r * (size_t)W + (size_t)c);
                const float ex = e[0], ey = e[1], ez = e[2];
                const bool fin = isfinite(ex + (ey + ez));        // Eigen's Vector3f::sum(): x + (y + z)
                for (int a = 0; a < 3; ++a) {
                    out[a] = (prev_n[a] + cur_c[a]) - prev_c[a];
                    if (fin) out[a] += (double)(a == 0 ? ex : a == 1 ? ey : ez);
                    cur[3 * (size_t)(c + 1) + a] = out[a];
                    prev_c[a] = prev_n[a];
                    cur_c[a] = out[a];
                }
            }
            __syncthreads();
            if (active)
                for (int a = 0; a < 3; ++a) ring[j][c & 1][a] = out[a];
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
    }
}

__device__ __forceinline__ void ii_sum(const double *ii, int W, int sx, int sy, int w, int h, double out[3]) {
    const size_t ul = (size_t)sy * (size_t)(W + 1) + (size_t)sx, ur = ul + (size_t)w;
    const size_t ll = (size_t)(sy + h) * (size_t)(W + 1) + (size_t)sx, lr = ll + (size_t)w;
    for (int a = 0; a < 3; ++a) out[a] = ((ii[3 * lr + a] + ii[3 * ul + a]) - ii[3 * ur + a]) - ii[3 * ll + a];
}

// computeFeatureFull (BORDER_POLICY_IGNORE, smoothing independent of depth) + computePointNormal
__global__ __launch_bounds__(256) void on_normals_kernel(OrganizedView v) {
    const int border = (int)v.smoothing;
    const int ci = border + blockIdx.x * blockDim.x + threadIdx.x, ri = border + blockIdx.y;
    if (ci >= v.W - border || ri >= v.H - border) return;
    const size_t index = (size_t)ri * (size_t)v.W + (size_t)ci;
    const float *pt = pixel(v, index);
    const float px = pt[0], py = pt[1], pz = pt[2];
    if (!isfinite(pz)) return;
    const float d = v.dist[index];
    const float smoothing = d < v.smoothing ? d : v.smoothing;
    if (!(smoothing > 2.0f)) return;
    const int rw = (int)smoothing, rh = (int)smoothing, rw2 = rw / 2, rh2 = rh / 2;
    double s1[3], s0[3], gx[3], gy[3];
    ii_sum(v.ii, v.W, ci + rw2, ri - rh2, 1, rh, s1);
    ii_sum(v.ii, v.W, ci - rw2, ri - rh2, 1, rh, s0);
    for (int a = 0; a < 3; ++a) gx[a] = s1[a] - s0[a];
    ii_sum(v.ii, v.W, ci - rw2, ri + rh2, rw, 1, s1);
    ii_sum(v.ii, v.W, ci - rw2, ri - rh2, rw, 1, s0);
    for (int a = 0; a < 3; ++a) gy[a] = s1[a] - s0[a];
    const double nv[3] = {gy[1] * gx[2] - gy[2] * gx[1], gy[2] * gx[0] - gy[0] * gx[2], gy[0] * gx[1] - gy[1] * gx[0]};
    const double len = (nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2];
    if (len == 0.0) return;
    const double root = sqrt(len);
    float nx = (float)(nv[0] / root), ny = (float)(nv[1] / root), nz = (float)(nv[2] / root);
    const float vx = v.vp[0] - px, vy = v.vp[1] - py, vz = v.vp[2] - pz;       // pcl::flipNormalTowardsViewpoint
    const float cos_theta = (vx * nx + vy * ny + vz * nz);
    if (cos_theta < 0) {
        nx *= -1;
        ny *= -1;
        nz *= -1;
    }
    float *o = reinterpret_cast<float *>(v.normals + index * v.ns);
    o[0] = nx;
    o[1] = ny;
    o[2] = nz;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

void preload_organized_normals_code() {        // (kernels.hip, preload_code)
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&on_init_kernel));
}

size_t organized_normals_scratch_bytes(int W, int H) {
    const size_t n = (size_t)(W > 0 ? W : 1) * (size_t)(H > 0 ? H : 1);
    return align256(n) + align256(sizeof(float) * n) + align256(sizeof(double) * 3 * (size_t)(W + 1) * (size_t)(H + 1));
}

void launch_organized_normals(OrganizedView v, void *scratch, hipStream_t st) {
    const size_t n = (size_t)v.W * (size_t)v.H;
    if (v.W <= 0 || v.H <= 0) return;
    char *p = static_cast<char *>(scratch);
    v.change = reinterpret_cast<unsigned char *>(p);
    p += align256(n);
    v.dist = reinterpret_cast<float *>(p);
    p += align256(sizeof(float) * n);
    v.ii = reinterpret_cast<double *>(p);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    // row 0 of the integral image (3 (W + 1) doubles) is cleared by the same launch: the grid covers both extents
    const size_t row0 = (size_t)3 * (size_t)(v.W + 1);
    on_init_kernel<<<(unsigned)(((n > row0 ? n : row0) + 255) / 256), 256, 0, st>>>(v);
    const int border = (int)v.smoothing;
    if (border < 0 || v.W <= 2 * border || v.H <= 2 * border) return;       // nothing but NaN
    on_change_kernel<<<dim3((unsigned)((v.W + 255) / 256), (unsigned)v.H), 256, 0, st>>>(v);
    on_dist_init_kernel<<<blocks, 256, 0, st>>>(v);
    on_dist_pass_kernel<true><<<1, kRows, 0, st>>>(v);
    on_dist_pass_kernel<false><<<1, kRows, 0, st>>>(v);
    on_integral_kernel<<<1, kRows, 0, st>>>(v);
    on_normals_kernel<<<dim3((unsigned)((v.W - 2 * border + 255) / 256), (unsigned)(v.H - 2 * border)), 256, 0, st>>>(v);
}

}  // namespace kpl
