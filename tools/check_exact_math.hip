// check_exact_math.hip -- the two exact-arithmetic shortcuts of csrc/exact_math.h against hipcc's own IEEE operations,
// ON THE DEVICE.  Run by tests/test_gpu_exact_math.py; prints one JSON line per mode.
//
//   (no argument) / "sqrt"   sqrt_rn against sqrtf for EVERY float of its domain: +0 and 2^-96 <= x < 2^127 (1.87e9
//                            values), plus the values below 2^-96 through the wave-uniform fallback.
//   "div"                    div_rn(a, b, RN(1/b)) against a / b for >= 64 divisors b -- the per-launch constants of the
//                            soft assignment are arbitrary: support / A for a caller's radius, 2 / B --: random r / A,
//                            2 / (float)B for B = 1 .. 32, and mantissa edge cases (0x000000, 0x7fffff, 0x400000 and their
//                            neighbours) at several exponents; for each, EVERY numerator in [2^-100, 16 b] (value / dim)
//                            and in [-b, -2^-100] ((value - center) / dim lies in [-0.5, 0.5]).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

// every float whose bit pattern lies in [lo, hi] (one sign) as numerator of divisor b: div_rn against hipcc's division
__global__ void check_div(float b, float rb, unsigned lo, unsigned hi, unsigned long long *bad, unsigned *example) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (unsigned long long u = (unsigned long long)lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u <= hi; u += stride) {
        const float a = __uint_as_float((unsigned)u);
        const float got = kpl::div_rn(a, b, rb), want = a / b;
        if (__float_as_uint(got) != __float_as_uint(want)) {
            ++mine;
            example[0] = (unsigned)u;
            example[1] = __float_as_uint(b);
        }
    }
    if (mine) atomicAdd(bad, mine);
}

static float from_bits(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint32_t to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static int run_div() {
    std::vector<float> divisors;
    // 2 / (float)B, the bin dimension (cpp:75), B = 1 .. 32
    for (int B = 1; B <= 32; ++B) divisors.push_back(2 / (float)B);
    // support / A for random radii (cpp:43), A = 1 .. 16: a fixed 64-bit generator, no library distribution
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto next = [&]() {
        s += 0x9e3779b97f4a7c15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    for (int k = 0; k < 48; ++k) {
        const float r = ldexpf(1.0f + (float)(next() >> 40) / 16777216.0f, (int)(next() % 14) - 7);   // 2^-7 .. 2^7
        const int A = 1 + (int)(next() % 16);
        divisors.push_back(r / (float)A);
    }
    // mantissa edge cases
    const uint32_t mants[] = {0x000000, 0x000001, 0x3fffff, 0x400000, 0x400001, 0x7ffffe, 0x7fffff, 0x555555, 0x2aaaaa};
    for (uint32_t m : mants)
        for (int e = 118; e <= 134; e += 8) divisors.push_back(from_bits(((uint32_t)e << 23) | m));
    unsigned long long *d_bad, bad = 0, checked = 0;
    unsigned *d_ex, ex[2] = {0, 0};
    if (hipMalloc(&d_bad, 8) != hipSuccess || hipMalloc(&d_ex, 8) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    (void)hipMemset(d_bad, 0, 8);
    (void)hipMemset(d_ex, 0, 8);
    for (float b : divisors) {
        const float rb = 1.0f / b;                        // RN(1 / b), as api.cpp's make_feat computes it on the host
        const unsigned plo = to_bits(ldexpf(1.0f, -100)), phi = to_bits(16.0f * b);
        const unsigned nlo = to_bits(-ldexpf(1.0f, -100)), nhi = to_bits(-b);
        check_div<<<2048, 256>>>(b, rb, plo, phi, d_bad, d_ex);
        check_div<<<2048, 256>>>(b, rb, nlo, nhi, d_bad, d_ex);
        checked += (unsigned long long)(phi - plo + 1) + (unsigned long long)(nhi - nlo + 1);
    }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ex, d_ex, 8, hipMemcpyDeviceToHost);
    printf("{\"mode\": \"div\", \"divisors\": %zu, \"checked\": %llu, \"mismatches\": %llu, \"example_numerator_bits\": %u, "
           "\"example_divisor_bits\": %u}\n", divisors.size(), checked, bad, ex[0], ex[1]);
    return bad == 0 ? 0 : 3;
}

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "div") == 0) return run_div();
    unsigned long long *d_bad, bad = 0;
    unsigned *d_ex, ex = 0;
    if (hipMalloc(&d_bad, 8) != hipSuccess || hipMalloc(&d_ex, 4) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    (void)hipMemset(d_bad, 0, 8);
    (void)hipMemset(d_ex, 0, 4);
    // all non-negative finite floats: bit patterns 0 .. 0x7f7fffff (the values below 2^-96 take the fallback)
    const unsigned long long count = 0x7f800000ull;
    check<<<4096, 256>>>(0u, count, d_bad, d_ex);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&ex, d_ex, 4, hipMemcpyDeviceToHost);
    printf("{\"checked\": %llu, \"mismatches\": %llu, \"example_bits\": %u}\n", count, bad, ex);
    return bad == 0 ? 0 : 3;
}
