// valu_ceiling.hip -- measures the VALU issue ceiling of one gfx950 SIMD: wave-instructions per shader cycle,
// for the instruction kinds the feature kernel is made of, at 1 / 2 / 4 (/ 8) resident waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_ceiling tools/valu_ceiling.hip && tools/valu_ceiling
//
// Every CU gets ONE workgroup of 256 * W threads (W waves on each of its 4 SIMDs; 96 KB of LDS per workgroup
// keep a second one off the CU; W = 8 uses two workgroups of 1024 threads with 64 KB each).  A wave runs
// ITERS x 64 instructions of one kind on 8 independent registers (no dependency within 8 instructions) between two
// s_memtime stamps (shader cycles).  Reported per kind and W:
//   ipc = W x ITERS x 64 / median over waves of (t1 - t0)      wave-instructions per cycle and SIMD
// i.e. the ceiling that SQ_INSTS_VALU / (SIMDs x kernel cycles) of a real kernel has to be compared with.
// The output (one JSON line) is kept under profiles/ and read by bench.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int kIters = 4096;

// 8 instructions on 8 different registers; the source operand b never changes
#define OP8(INS)                                                                                   \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                            \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(b), "v"(ib))
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %8\n\t"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n\t"
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n\t"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n\t"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n\t"
#define I_FLOOR(k) "v_floor_f32 %" #k ", %" #k "\n\t"
#define I_CVT(k) "v_cvt_i32_f32 %" #k ", %" #k "\n\t"
#define I_ALIGN(k) "v_alignbit_b32 %" #k ", %" #k ", %9, 31\n\t"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define I_MUL24(k) "v_mul_i32_i24 %" #k ", %" #k ", %9\n\t"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %9\n\t"
#define I_DPP(k) "v_or_b32_dpp %" #k ", %" #k ", %" #k " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\t"

enum Kind { ADD, MUL, FMA, SQRT, RCP, FLOOR, CVT, ALIGN, CNDMASK, MUL24, ADDU, DPP, CMP, MIX, NKINDS };
static const char *kNames[NKINDS] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_sqrt_f32", "v_rcp_f32", "v_floor_f32",
                                     "v_cvt_i32_f32", "v_alignbit_b32", "v_cndmask_b32", "v_mul_i32_i24", "v_add_u32",
                                     "v_or_b32_dpp", "v_cmp_lt_f32", "mix_feature_drain"};

template <int KIND>
__global__ __launch_bounds__(1024) void chain(float *sink, unsigned long long *cyc, float seed) {
    extern __shared__ float pad[];
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f;
    const int ib = 3;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == ADD) OP8(I_ADD);
            else if (KIND == MUL) OP8(I_MUL);
            else if (KIND == FMA) OP8(I_FMA);
            else if (KIND == SQRT) OP8(I_SQRT);
            else if (KIND == RCP) OP8(I_RCP);
            else if (KIND == FLOOR) OP8(I_FLOOR);
            else if (KIND == CVT) OP8(I_CVT);
            else if (KIND == ALIGN) OP8(I_ALIGN);
            else if (KIND == CNDMASK) OP8(I_CNDMASK);
            else if (KIND == MUL24) OP8(I_MUL24);
            else if (KIND == ADDU) OP8(I_ADDU);
            else if (KIND == DPP) OP8(I_DPP);
            else if (KIND == CMP) asm volatile(I_CMP(0) I_CMP(1) I_CMP(2) I_CMP(3) I_CMP(4) I_CMP(5) I_CMP(6) I_CMP(7)
                                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                               : "v"(b), "v"(ib)
                                               : "vcc");
            else if (u == 0) {
                // the instruction mix of one drain round of the feature kernel (~114 VALU instructions per accepted
                // neighbor and lane): per 128 instructions 1 sqrt, 4 fma (the two exact divisions), 2 floor, 1 cvt,
                // 32 float add / mul, 24 selects, 64 integer / address instructions
                asm volatile(I_SQRT(0) I_FMA(1) I_FMA(2) I_FMA(3) I_FMA(4) I_FLOOR(5) I_FLOOR(6) I_CVT(7)
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(ib));
                OP8(I_ADD); OP8(I_MUL); OP8(I_ADD); OP8(I_MUL);
                OP8(I_CNDMASK); OP8(I_CNDMASK); OP8(I_CNDMASK);
                OP8(I_ADDU); OP8(I_MUL24); OP8(I_ALIGN); OP8(I_ADDU); OP8(I_ADDU); OP8(I_MUL24); OP8(I_ALIGN); OP8(I_ADDU);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f) sink[0] = s + pad[threadIdx.x];
}

template <int KIND>
static double run(int W, int cus, float *sink, unsigned long long *d_cyc, double *ms_out) {
    const int blocks_per_cu = W > 4 ? 2 : 1, waves_per_block = 4 * W / blocks_per_cu;
    const size_t lds = W > 4 ? 64 * 1024 : 96 * 1024;
    const int nblocks = cus * blocks_per_cu, nwaves = nblocks * waves_per_block;
    CHECK(hipFuncSetAttribute((const void *)chain<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int warm = 0; warm < 2; ++warm) chain<KIND><<<nblocks, waves_per_block * 64, lds>>>(sink, d_cyc, 1.0f);
    CHECK(hipEventRecord(e0));
    chain<KIND><<<nblocks, waves_per_block * 64, lds>>>(sink, d_cyc, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms;
    std::vector<unsigned long long> c(nwaves);
    CHECK(hipMemcpy(c.data(), d_cyc, sizeof(unsigned long long) * nwaves, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double med = (double)c[nwaves / 2];
    const double per_iter = KIND == MIX ? 128.0 : 64.0;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return (double)W * kIters * per_iter / med;
}

int main() {
    int dev = 0, cus = 0, clk = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
    float *sink;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * (size_t)cus * 64));
    const int Ws[4] = {1, 2, 4, 8};
    printf("{\"device_cus\": %d, \"clock_khz\": %d, \"unit\": \"wave-instructions per shader cycle per SIMD (s_memtime)\", "
           "\"iters\": %d, \"kinds\": {", cus, clk, kIters);
    for (int k = 0; k < NKINDS; ++k) {
        printf("%s\"%s\": {", k ? ", " : "", kNames[k]);
        for (int wi = 0; wi < 4; ++wi) {
            const int W = Ws[wi];
            double ms = 0, ipc = 0;
            switch (k) {
#define CASE(K) case K: ipc = run<K>(W, cus, sink, d_cyc, &ms); break;
                CASE(ADD) CASE(MUL) CASE(FMA) CASE(SQRT) CASE(RCP) CASE(FLOOR) CASE(CVT) CASE(ALIGN) CASE(CNDMASK)
                CASE(MUL24) CASE(ADDU) CASE(DPP) CASE(CMP) CASE(MIX)
#undef CASE
            }
            printf("%s\"w%d\": {\"ipc\": %.4f, \"cycles_per_instr\": %.3f, \"kernel_ms\": %.4f}", wi ? ", " : "", W, ipc, 1.0 / ipc, ms);
        }
        printf("}");
    }
    printf("}}\n");
    return 0;
}
