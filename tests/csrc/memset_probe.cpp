// memset_probe.cpp -- is hipMemset (null stream, device memory) complete when it returns, and is it ordered before work
// that the same host thread enqueues afterwards on a hipStreamNonBlocking stream?
//
// libkpl clears freshly grown tables with hipMemset and then launches kernels on the handle's NON-BLOCKING stream
// (api.cpp: ensure_cells, prepare_detect).  CUDA documents cudaMemset as asynchronous with respect to the host for
// device memory, and a non-blocking stream does not synchronise with the null stream: if HIP behaves the same, a kernel
// of that stream can write into the table BEFORE the memset has passed over it -- and loses its writes.
// The probe: hipMemset a large buffer to 0, immediately launch a kernel on a non-blocking stream that writes a marker
// into every 4096th int, synchronise everything, count the markers that survived; and time the hipMemset call itself.
//   hipcc -O2 -o memset_probe memset_probe.cpp ; ./memset_probe [MiB] [iterations]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            return 2;                                                                         \
        }                                                                                     \
    } while (0)

__global__ void mark(int *p, size_t n_marks, int value) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_marks) p[i * 4096] = value;
}

__global__ void count_marks(const int *p, size_t n_marks, int value, unsigned long long *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_marks && p[i * 4096] == value) atomicAdd(out, 1ull);
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const int iters = argc > 2 ? atoi(argv[2]) : 50;
    const size_t bytes = mib << 20, n_int = bytes / 4, n_marks = n_int / 4096;
    int *buf = nullptr;
    unsigned long long *d_cnt = nullptr, h_cnt = 0;
    CHECK(hipMalloc((void **)&buf, bytes));
    CHECK(hipMalloc((void **)&d_cnt, 8));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    long lost_total = 0;
    double call_us = 0.0;
    for (int it = 0; it < iters; ++it) {
        CHECK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(hipMemset(buf, 0, bytes));                                     // null stream
        const auto t1 = std::chrono::steady_clock::now();
        mark<<<(unsigned)((n_marks + 255) / 256), 256, 0, st>>>(buf, n_marks, it + 1);          // non-blocking stream, at once
        CHECK(hipStreamSynchronize(st));
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(d_cnt, 0, 8));
        CHECK(hipDeviceSynchronize());
        count_marks<<<(unsigned)((n_marks + 255) / 256), 256, 0, st>>>(buf, n_marks, it + 1, d_cnt);
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(&h_cnt, d_cnt, 8, hipMemcpyDeviceToHost));
        lost_total += (long)(n_marks - h_cnt);
        call_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
    }
    printf("%zu MiB, %d iterations: hipMemset call %.1f us on average (%.1f GB/s if it were complete on return); "
           "markers written right after it on a non-blocking stream and lost: %ld of %zu\n",
           mib, iters, call_us / iters, bytes / (call_us / iters) / 1e3, lost_total, n_marks * (size_t)iters);
    printf(lost_total ? "PROBE: the memset is NOT ordered before the other stream's kernel\n" : "PROBE: no marker lost\n");
    return lost_total ? 1 : 0;
}
