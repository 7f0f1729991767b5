// host_cost_probe.cpp -- what the FIRST use of a stream / copy direction / kernel costs the calling thread on this runtime
// (ROCm 7.2): the numbers behind kpl_create's eager set-up ("the first call").  Stand-alone; not part of libkpl.
//   hipcc --offload-arch=gfx950 -O2 -o host_cost_probe host_cost_probe.cpp && ./host_cost_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void touch(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void touch2(int *p) { if (threadIdx.x == 1 && blockIdx.x == 0) p[1] += 1; }

__global__ void spin(int *p, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) p[2] += 1;
}
static hipError_t launch_spin(hipStream_t st, int *d) { spin<<<1, 64, 0, st>>>(d, 200000000ll / 100); return hipGetLastError(); }
static hipError_t launch1(hipStream_t st, int *d) { touch<<<1, 64, 0, st>>>(d); return hipGetLastError(); }
static hipError_t launch2(hipStream_t st, int *d) { touch2<<<1, 64, 0, st>>>(d); return hipGetLastError(); }

static double now_ms() {
    static auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define T(label, stmt)                                                      \
    do {                                                                    \
        const double a_ = now_ms();                                         \
        hipError_t e_ = (stmt);                                             \
        printf("%-64s %8.3f ms%s\n", label, now_ms() - a_, e_ == hipSuccess ? "" : "  FAILED"); \
    } while (0)

int main() {
    const size_t bytes = 1 << 20;
    std::vector<char> pageable(bytes, 1);
    int *d = nullptr, *pinned = nullptr;
    void *big = nullptr;
    T("hipSetDevice(0) (runtime init)", hipSetDevice(0));
    T("hipMalloc 1 MiB", hipMalloc(&big, bytes));
    T("hipMalloc 64 B", hipMalloc(&d, 64));
    T("hipHostMalloc 1 MiB", hipHostMalloc((void **)&pinned, bytes, hipHostMallocDefault));
    T("hipMemset (null stream)", hipMemset(d, 0, 64));
    T("hipDeviceSynchronize", hipDeviceSynchronize());
    for (int s = 0; s < 4; ++s) {
        printf("--- stream %d\n", s);
        hipStream_t st;
        T("hipStreamCreateWithFlags(NonBlocking)", hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        if (s == 1) {        // variant: the stream's first operation is a kernel, not a copy
            T("first kernel launch on it", launch1(st, d));
            T("hipStreamSynchronize", hipStreamSynchronize(st));
        }
        T("hipMemcpyAsync H2D 1 MiB pageable", hipMemcpyAsync(big, pageable.data(), bytes, hipMemcpyHostToDevice, st));
        T("hipMemcpyAsync H2D 1 MiB pageable (again)", hipMemcpyAsync(big, pageable.data(), bytes, hipMemcpyHostToDevice, st));
        T("kernel launch", launch1(st, d));
        T("second kernel (first launch of it only on stream 0)", launch2(st, d));
        T("hipMemcpyAsync D2H 4 B into pinned", hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        T("hipMemcpyAsync D2H 4 B into pinned (again)", hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        T("spin kernel (~1 ms) launch", launch_spin(st, d));
        T("hipMemcpyAsync D2H 4 B into pinned BEHIND the running kernel", hipMemcpyAsync(pinned, d, 4, hipMemcpyDeviceToHost, st));
        T("hipMemcpyAsync D2H 128 KiB into pinned BEHIND the running kernel", hipMemcpyAsync(pinned, big, 128 << 10, hipMemcpyDeviceToHost, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        T("spin kernel (~1 ms) launch", launch_spin(st, d));
        T("hipMemcpyAsync D2H 128 KiB into pinned BEHIND the running kernel (again)", hipMemcpyAsync(pinned, big, 128 << 10, hipMemcpyDeviceToHost, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        T("hipMemcpyAsync D2H 128 KiB into pinned", hipMemcpyAsync(pinned, big, 128 << 10, hipMemcpyDeviceToHost, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        T("hipMemcpyAsync D2H 128 KiB into pinned (again)", hipMemcpyAsync(pinned, big, 128 << 10, hipMemcpyDeviceToHost, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        T("hipMemcpyAsync H2D 1 MiB from pinned", hipMemcpyAsync(big, pinned, bytes, hipMemcpyHostToDevice, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        T("hipMemcpyAsync H2D 1 MiB from pinned (again)", hipMemcpyAsync(big, pinned, bytes, hipMemcpyHostToDevice, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
        void *tmp = nullptr;
        T("hipMallocAsync 1 MiB", hipMallocAsync(&tmp, bytes, st));
        hipEvent_t ev;
        T("hipEventCreateWithFlags", hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        T("hipEventRecord", hipEventRecord(ev, st));
        T("hipStreamSynchronize", hipStreamSynchronize(st));
    }
    printf("pinned[0] = %d\n", pinned[0]);
    return 0;
}
