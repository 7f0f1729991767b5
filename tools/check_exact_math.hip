// check_exact_math.hip -- sqrt_rn (csrc/exact_math.h) against hipcc's correctly rounded sqrtf for EVERY float
// of its domain: +0 and 2^-96 <= x < 2^127 (1.87e9 values), plus the values below 2^-96 through the wave-uniform
// fallback.  Prints one JSON line: {"checked": N, "mismatches": M, ...}.  Run by tests/test_gpu_exact_math.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "exact_math.h"

__global__ void check(unsigned first, unsigned long long count, unsigned long long *bad, unsigned *example) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count + (stride - count % stride) % stride; i += stride) {
        const bool in = i < count;                       // whole waves stay in the loop: sqrt_rn votes across the wave
        const unsigned bits = first + (unsigned)(in ? i : 0);
        const float x = __uint_as_float(bits);
        const float a = kpl::sqrt_rn(x), b = sqrtf(x);
        if (in && __float_as_uint(a) != __float_as_uint(b)) {
            ++mine;
            *example = bits;
        }
    }
    if (mine) atomicAdd(bad, mine);
}

int main() {
    unsigned long long *d_bad, bad = 0;
    unsigned *d_ex, ex = 0;
    if (hipMalloc(&d_bad, 8) != hipSuccess || hipMalloc(&d_ex, 4) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    hipMemset(d_bad, 0, 8);
    hipMemset(d_ex, 0, 4);
    // all non-negative finite floats: bit patterns 0 .. 0x7f7fffff (the values below 2^-96 take the fallback)
    const unsigned long long count = 0x7f800000ull;
    check<<<4096, 256>>>(0u, count, d_bad, d_ex);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&ex, d_ex, 4, hipMemcpyDeviceToHost);
    printf("{\"checked\": %llu, \"mismatches\": %llu, \"example_bits\": %u}\n", count, bad, ex);
    return bad == 0 ? 0 : 3;
}
