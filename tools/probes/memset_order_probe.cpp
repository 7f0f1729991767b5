// memset_order_probe.cpp -- hunting the "fresh handle counts nothing" event of round 6 outside libkpl (ROCm 7.2, MI355X).
// Mimics a libkpl handle's life: stream, [set-up: pool block allocated, CLEARED, freed, stream synced], first call = ~20
// stream-ordered allocations (three of them cleared), a kernel that sets every flag, a kernel that counts them; then the
// handle is destroyed the way kpl_destroy does it (device sync, hipFreeAsync on the null stream, stream destroyed).
//   hipcc --offload-arch=gfx950 -O2 -o memset_order_probe memset_order_probe.cpp && ./memset_order_probe [warm: 0|1|2]
//   warm 0: no set-up block; 1: allocated + cleared + freed (what failed in libkpl); 2: allocated + freed, not cleared
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void set_ones(int *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1; }
__global__ void touch(int *p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }
__global__ void count_ones(const int *p, int n, int *out) {
    int c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) c += p[i] != 0;
    atomicAdd(out, c);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int warm = argc > 1 ? atoi(argv[1]) : 1;
    CK(hipSetDevice(0));
    int *out = nullptr, *h_out = nullptr;
    CK(hipMalloc(&out, 4));
    CK(hipHostMalloc((void **)&h_out, 4, hipHostMallocDefault));
    const size_t sizes[] = {391946, 8256, 64256, 31006, 371, 32336, 32256, 8256, 8256, 240256, 8261, 261, 8256, 8266, 276};
    const int nsz = sizeof(sizes) / sizeof(sizes[0]), kFlags = 10, kCand = 11, kScan = 14;
    const int n = 1600;
    int bad = 0, rounds = 0;
    for (int rep = 0; rep < 60; ++rep) {
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        if (warm) {
            void *tmp = nullptr;
            CK(hipMallocAsync(&tmp, 128 << 10, st));
            if (warm == 1) CK(hipMemsetAsync(tmp, 0, 128 << 10, st));
            CK(hipFreeAsync(tmp, st));
            CK(hipStreamSynchronize(st));
        }
        std::vector<void *> arr(nsz, nullptr);
        for (int k = 0; k < nsz; ++k) {
            CK(hipMallocAsync(&arr[k], sizes[k], st));
            if (k == 0 || k == kFlags || k == kCand || k == kScan) CK(hipMemsetAsync(arr[k], 0, sizes[k], st));
        }
        CK(hipMemsetAsync(out, 0, 4, st));
        for (int k = 1; k < 8; ++k) touch<<<8, 256, 0, st>>>((int *)arr[k], (int)(sizes[k] / 4));
        set_ones<<<(n + 255) / 256, 256, 0, st>>>((int *)arr[kFlags], n);
        // ... and the array that took the place of the set-up block: written right after its own clear, counted at the end
        const int n0 = (int)(sizes[0] / 4);
        set_ones<<<(n0 + 255) / 256, 256, 0, st>>>((int *)arr[0], n0);
        for (int k = 1; k < 8; ++k) touch<<<8, 256, 0, st>>>((int *)arr[k], (int)(sizes[k] / 4));
        count_ones<<<1, 256, 0, st>>>((const int *)arr[0], n0, out);
        count_ones<<<1, 256, 0, st>>>((const int *)arr[kFlags], n, out);
        CK(hipMemcpyAsync(h_out, out, 4, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        ++rounds;
        if (*h_out != n + n0) { ++bad; printf("rep %d: counted %d of %d\n", rep, *h_out, n + n0); }
        CK(hipDeviceSynchronize());
        for (int k = 0; k < nsz; ++k) CK(hipFreeAsync(arr[k], nullptr));
        CK(hipStreamDestroy(st));
    }
    printf("warm=%d: %d of %d handle lives lost their flags\n", warm, bad, rounds);
    return 0;
}
