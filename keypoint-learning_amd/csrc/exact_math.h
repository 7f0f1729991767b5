// exact_math.h -- correctly rounded float division and square root in fewer instructions than hipcc's
// expansions, for the inner loop of the feature kernels (kernels.hip).  Both return exactly the IEEE result the
// reference computes (src/KeypointLearning.cpp:45,55,76,85 divide; impl/KeypointLearning.hpp:345 sqrt).
// Also included by tools/check_exact_math.hip, which compares sqrt_rn with sqrtf for EVERY float of its domain
// on the device (tests/test_gpu_exact_math.py); div_rn is checked on the CPU (tests/test_division.py).
#pragma once
#include <hip/hip_runtime.h>

namespace kpl {

// div_rn(a, b, rb) returns the correctly rounded float quotient a / b in 3 instructions instead of hipcc's
// ~11-instruction expansion: with rb = RN(1/b), q = RN(a*rb) is within one ulp of a/b, r = a - q*b is exact in
// one FMA, and RN(q + r*rb) is the correctly rounded quotient (Markstein's division theorem; b is a positive
// normal constant here and a/b stays far from overflow; a quotient that underflows is only ever floored to 0).
// The FMAs are explicit, -ffp-contract=off stays in force for everything else.
__device__ __forceinline__ float div_rn(float a, float b, float rb) {
    const float q = a * rb;
    const float r = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(r, rb, q);
}

// sqrt_rn(x) = sqrtf(x), correctly rounded, for x = +0 or 2^-96 <= x < 2^127 (squared distances below r^2), in 9
// instructions instead of hipcc's 16: v_sqrt_f32 is within one ulp, so the result is s, its predecessor or its
// successor, decided by the signs of the exact residuals x - sd*s and x - su*s (one FMA each) -- hipcc's own
// algorithm without the rescaling of tiny arguments (their residuals would underflow) and without the special case
// of zero / infinity (zero comes out right: the residuals are NaN and 0, neither test fires).  Tiny arguments
// take hipcc's sqrtf behind a wave-uniform branch that is practically never taken.
constexpr float kSqrtRnMin = 0x1.0p-96f;
__device__ __forceinline__ float sqrt_rn_core(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}
__device__ __forceinline__ float sqrt_rn(float x) {
    if (__builtin_expect(__any((x < kSqrtRnMin) & (x > 0.0f)), 0)) return sqrtf(x);
    return sqrt_rn_core(x);
}

}  // namespace kpl
