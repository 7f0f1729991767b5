/* Exhaustive check of the 3-instruction division used in the HIP kernels
 * (keypoint-learning_amd/csrc/kernels.hip div_rn) against IEEE division, on the CPU:
 * q = a*rb; r = fma(-q, b, a); result = fma(r, rb, q) with rb = 1.0f / b. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float div_rn(float a, float b, float rb)
{
    float q = a * rb;
    float r = fmaf(-q, b, a);
    return fmaf(r, rb, q);
}

/* all floats whose bit pattern lies in [lo, hi) (same sign), stepping `step`; returns mismatches
 * and stores the first offending pattern */
long check_range(float b, uint32_t lo, uint32_t hi, uint32_t step, uint32_t *first_bad)
{
    const float rb = 1.0f / b;
    long bad = 0;
    for (uint64_t u = lo; u < hi; u += step) {
        uint32_t bits = (uint32_t)u;
        float a;
        memcpy(&a, &bits, 4);
        float want = a / b, got = div_rn(a, b, rb);
        if (memcmp(&want, &got, 4) != 0 && !(want != want && got != got)) {
            if (!bad && first_bad) *first_bad = bits;
            ++bad;
        }
    }
    return bad;
}
