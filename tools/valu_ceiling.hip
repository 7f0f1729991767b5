// valu_ceiling.hip -- measures the VALU issue ceiling of one gfx950 SIMD: wave-instructions per shader cycle,
// for the instruction kinds the feature kernel is made of, at 1 / 2 / 4 (/ 8) resident waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_ceiling tools/valu_ceiling.hip && tools/valu_ceiling
//
// Every CU gets ONE workgroup of 256 * W threads (W waves on each of its 4 SIMDs; 96 KB of LDS per workgroup
// keep a second one off the CU; W = 8 uses two workgroups of 1024 threads with 64 KB each).  A wave runs
// ITERS x 64 instructions of one kind on 8 independent registers (no dependency within 8 instructions) between two
// s_memtime stamps (shader cycles).  Reported per kind and W:
//   ipc      = W x ITERS x 64 / median over waves of (t1 - t0)      wave-instructions per cycle and SIMD
//   ipc_wall = W x ITERS x 64 / (kernel time by HIP events x in-kernel clock); the clock is the median over waves
//              of delta s_memtime / delta s_memrealtime x 100 MHz.  The two agree when the W waves of a SIMD are
//              resident together for the whole kernel (W <= 4: one workgroup per CU); W = 8 needs two workgroups
//              per CU, which the dispatcher may run one after the other: trust ipc_wall there.
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
#define I_CNDS(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, %10\n\t"   /* condition in an SGPR pair */
#define I_BFI(k) "v_bfi_b32 %" #k ", %9, %" #k ", %8\n\t"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n\t"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %9\n\t"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n\t"
#define I_SUB(k) "v_sub_f32 %" #k ", %" #k ", %8\n\t"
#define I_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %9, %9\n\t"
#define I_FFBH(k) "v_ffbh_u32 %" #k ", %" #k "\n\t"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n\t"
#define I_OR(k) "v_or_b32 %" #k ", %" #k ", %9\n\t"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %9\n\t"
#define I_SUBU(k) "v_sub_u32 %" #k ", %" #k ", %9\n\t"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %8, %8\n\t"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %9\n\t"
#define I_MINI(k) "v_min_i32 %" #k ", %" #k ", %9\n\t"
#define I_CMPCND(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n\tv_cndmask_b32 %" #k ", %" #k ", %8, vcc\n\t"
#define OP8S(INS)                                                                                  \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                            \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(b), "v"(ib), "s"(smask))

enum Kind { ADD, MUL, FMA, SQRT, RCP, FLOOR, CVT, ALIGN, CNDMASK, MUL24, ADDU, DPP, CMP, CNDS, BFI, MAX, AND, LSHL, SUB,
            MAD24, FFBH, MOV, OR, XOR, SUBU, FMAC, LSHLADD, MINI, CMPCND, PKADD, PKMUL, PKFMA, LDSR, LDSW, MIX, NKINDS };
static const char *kNames[NKINDS] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_sqrt_f32", "v_rcp_f32", "v_floor_f32",
                                     "v_cvt_i32_f32", "v_alignbit_b32", "v_cndmask_b32_vcc", "v_mul_i32_i24", "v_add_u32",
                                     "v_or_b32_dpp", "v_cmp_lt_f32", "v_cndmask_b32_sgpr", "v_bfi_b32", "v_max_f32",
                                     "v_and_b32", "v_lshlrev_b32", "v_sub_f32", "v_mad_u32_u24", "v_ffbh_u32", "v_mov_b32",
                                     "v_or_b32", "v_xor_b32", "v_sub_u32", "v_fmac_f32", "v_lshl_add_u32", "v_min_i32",
                                     "v_cmp_lt_f32+v_cndmask_b32_vcc (pair = 2 instructions)", "v_pk_add_f32", "v_pk_mul_f32",
                                     "v_pk_fma_f32", "ds_read_b32", "ds_write_b32", "mix_feature_drain"};

template <int KIND>
__global__ __launch_bounds__(1024) void chain(float *sink, unsigned long long *cyc, float seed, unsigned long long smask_in) {
    extern __shared__ float pad[];
    const unsigned long long smask = __builtin_amdgcn_readfirstlane((unsigned)smask_in) |
                                     ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(smask_in >> 32)) << 32);
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f;
    const int ib = 3;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const f2 pb = {b, b}, pc = {0.5f, 0.25f};
    const int laddr = (threadIdx.x & 63) * 4 + (threadIdx.x / 64) * 2048;      // 2 KB of LDS per wave, conflict-free
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
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
            else if (KIND == CNDS) OP8S(I_CNDS);
            else if (KIND == BFI) OP8(I_BFI);
            else if (KIND == MAX) OP8(I_MAX);
            else if (KIND == AND) OP8(I_AND);
            else if (KIND == LSHL) OP8(I_LSHL);
            else if (KIND == SUB) OP8(I_SUB);
            else if (KIND == MAD24) OP8(I_MAD24);
            else if (KIND == FFBH) OP8(I_FFBH);
            else if (KIND == MOV) OP8(I_MOV);
            else if (KIND == OR) OP8(I_OR);
            else if (KIND == XOR) OP8(I_XOR);
            else if (KIND == SUBU) OP8(I_SUBU);
            else if (KIND == FMAC) OP8(I_FMAC);
            else if (KIND == LSHLADD) OP8(I_LSHLADD);
            else if (KIND == MINI) OP8(I_MINI);
            else if (KIND == CMPCND) {
                if (u < 4) asm volatile(I_CMPCND(0) I_CMPCND(1) I_CMPCND(2) I_CMPCND(3) I_CMPCND(4) I_CMPCND(5) I_CMPCND(6) I_CMPCND(7)
                                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                        : "v"(b), "v"(ib)
                                        : "vcc");
            } else if (KIND == PKADD || KIND == PKMUL || KIND == PKFMA) {
                // 4 packed instructions on 4 register pairs (the same 8 registers)
                if (u < 8) {
                    if (KIND == PKADD) asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                                                    "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                                                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
                    else if (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                                                         "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                                                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
                    else asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                                      "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                                      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
                }
            } else if (KIND == LDSR) {
                asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
                             "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(laddr) : "memory");
            } else if (KIND == LDSW) {
                asm volatile("ds_write_b32 %8, %0\n\tds_write_b32 %8, %1 offset:256\n\tds_write_b32 %8, %2 offset:512\n\tds_write_b32 %8, %3 offset:768\n\t"
                             "ds_write_b32 %8, %4 offset:1024\n\tds_write_b32 %8, %5 offset:1280\n\tds_write_b32 %8, %6 offset:1536\n\tds_write_b32 %8, %7 offset:1792\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(laddr) : "memory");
            }
            else if (u == 0) {
                // the instruction mix of one drain round of the feature kernel (~114 VALU instructions per accepted
                // neighbor and lane): per 128 instructions 1 sqrt, 4 fma (the two exact divisions), 2 floor, 1 cvt,
                // 32 float add / mul, 24 selects, 64 integer / address instructions
                asm volatile(I_SQRT(0) I_FMA(1) I_FMA(2) I_FMA(3) I_FMA(4) I_FLOOR(5) I_FLOOR(6) I_CVT(7)
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(ib));
                OP8(I_ADD); OP8(I_MUL); OP8(I_ADD); OP8(I_MUL);
                OP8S(I_CNDS); OP8S(I_CNDS); OP8S(I_CNDS);
                OP8(I_ADDU); OP8(I_MUL24); OP8(I_ALIGN); OP8(I_ADDU); OP8(I_ADDU); OP8(I_MUL24); OP8(I_ALIGN); OP8(I_ADDU);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        cyc[2 * w] = t1 - t0;          // shader cycles
        cyc[2 * w + 1] = r1 - r0;      // ticks of the constant 100 MHz clock
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (s == 12345.678f) sink[0] = s + pad[threadIdx.x];
}

struct Result {
    double ipc_stamp;   // W x instructions / median wave's shader cycles (waves of a SIMD resident together)
    double ipc_wall;    // W x instructions / (kernel time x in-kernel clock)
    double ms, ghz;
};

template <int KIND>
static Result run(int W, int cus, float *sink, unsigned long long *d_cyc) {
    const int blocks_per_cu = W > 4 ? 2 : 1, waves_per_block = 4 * W / blocks_per_cu;
    const size_t lds = W > 4 ? 64 * 1024 : 96 * 1024;
    const int nblocks = cus * blocks_per_cu, nwaves = nblocks * waves_per_block;
    CHECK(hipFuncSetAttribute((const void *)chain<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const unsigned long long mask = 0x5555aaaa3333ccccull;
    for (int warm = 0; warm < 2; ++warm) chain<KIND><<<nblocks, waves_per_block * 64, lds>>>(sink, d_cyc, 1.0f, mask);
    CHECK(hipEventRecord(e0));
    chain<KIND><<<nblocks, waves_per_block * 64, lds>>>(sink, d_cyc, 1.0f, mask);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> raw(2 * (size_t)nwaves);
    CHECK(hipMemcpy(raw.data(), d_cyc, sizeof(unsigned long long) * raw.size(), hipMemcpyDeviceToHost));
    std::vector<double> c(nwaves), f(nwaves);
    for (int w = 0; w < nwaves; ++w) {
        c[w] = (double)raw[2 * w];
        f[w] = (double)raw[2 * w] / (double)raw[2 * w + 1] * 0.1;      // GHz: shader cycles per 10 ns tick
    }
    std::sort(c.begin(), c.end());
    std::sort(f.begin(), f.end());
    const double per_iter = KIND == MIX ? 128.0 : 64.0, instr = (double)W * kIters * per_iter;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    Result r;
    r.ms = ms;
    r.ghz = f[nwaves / 2];
    r.ipc_stamp = instr / c[nwaves / 2];
    r.ipc_wall = instr / (ms * 1e-3 * r.ghz * 1e9);
    return r;
}

int main() {
    int dev = 0, cus = 0, clk = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
    float *sink;
    unsigned long long *d_cyc;
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * (size_t)cus * 128));
    const int Ws[4] = {1, 2, 4, 8};
    printf("{\"device_cus\": %d, \"clock_khz\": %d, \"unit\": \"wave-instructions per shader cycle per SIMD (s_memtime)\", "
           "\"iters\": %d, \"kinds\": {", cus, clk, kIters);
    for (int k = 0; k < NKINDS; ++k) {
        printf("%s\"%s\": {", k ? ", " : "", kNames[k]);
        for (int wi = 0; wi < 4; ++wi) {
            const int W = Ws[wi];
            Result r{};
            switch (k) {
#define CASE(K) case K: r = run<K>(W, cus, sink, d_cyc); break;
                CASE(ADD) CASE(MUL) CASE(FMA) CASE(SQRT) CASE(RCP) CASE(FLOOR) CASE(CVT) CASE(ALIGN) CASE(CNDMASK)
                CASE(MUL24) CASE(ADDU) CASE(DPP) CASE(CMP) CASE(CNDS) CASE(BFI) CASE(MAX) CASE(AND) CASE(LSHL) CASE(SUB)
                CASE(MAD24) CASE(FFBH) CASE(MOV) CASE(OR) CASE(XOR) CASE(SUBU) CASE(FMAC) CASE(LSHLADD) CASE(MINI) CASE(CMPCND)
                CASE(PKADD) CASE(PKMUL) CASE(PKFMA) CASE(LDSR) CASE(LDSW) CASE(MIX)
#undef CASE
            }
            printf("%s\"w%d\": {\"ipc\": %.4f, \"ipc_wall\": %.4f, \"cycles_per_instr\": %.3f, \"kernel_ms\": %.4f, \"GHz\": %.3f}",
                   wi ? ", " : "", W, r.ipc_stamp, r.ipc_wall, 1.0 / r.ipc_stamp, r.ms, r.ghz);
        }
        printf("}");
    }
    printf("}}\n");
    return 0;
}
