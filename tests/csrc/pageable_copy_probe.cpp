// pageable_copy_probe.cpp -- does the HIP runtime move PAGEABLE host memory correctly when a host virtual address is
// unmapped and mapped again between two asynchronous copies (what malloc / free of a large numpy array does)?
//
// Round-3's one unexplained fuzz mismatch (tests/golden/fuzz_31337.npz) went through kpl_detect's host-buffer path:
// hipMemcpyAsync from and to pageable memory on the handle's non-blocking stream.  This probe exercises exactly that
// runtime path without libkpl: for several sizes around the runtime's staging / pinning thresholds it
//   maps a block at a FIXED address, fills it with a pattern of the iteration, copies it to the device on a
//   non-blocking stream, copies the device buffer back into pinned memory and compares; unmaps the block; repeats
//   (H2D), and the mirror image for D2H (device pattern -> a freshly mapped pageable block at the same address).
// Prints one line per size and direction: iterations, mismatches.  Exit code 1 on any mismatch.
//   hipcc -O2 -o pageable_copy_probe pageable_copy_probe.cpp ; ./pageable_copy_probe [iterations]
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                           \
            return 2;                                                                         \
        }                                                                                     \
    } while (0)

static void fill(uint32_t *p, size_t words, uint32_t seed) {
    uint32_t x = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < words; ++i) {
        x = x * 1664525u + 1013904223u;
        p[i] = x;
    }
}

static size_t count_diff(const uint32_t *a, const uint32_t *b, size_t words) {
    size_t d = 0;
    for (size_t i = 0; i < words; ++i) d += a[i] != b[i];
    return d;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const size_t sizes[] = {12 * 1024, 32 * 1024, 100 * 1024, 160 * 1024, 1 << 20, 5 << 20, 40 << 20};
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int bad_total = 0;
    for (size_t bytes : sizes) {
        const size_t words = bytes / 4;
        void *dev = nullptr, *pin = nullptr;
        CHECK(hipMalloc(&dev, bytes));
        CHECK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
        uint32_t *want = (uint32_t *)malloc(bytes);
        // a fixed address far from the heap
        void *const fixed = (void *)(uintptr_t)0x7e0000000000ull;
        long bad_h2d = 0, bad_d2h = 0;
        for (int it = 0; it < iters; ++it) {
            void *p = mmap(fixed, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0);
            if (p != fixed) {
                perror("mmap");
                return 2;
            }
            // ---- H2D from the freshly mapped block
            fill((uint32_t *)p, words, (uint32_t)(2 * it + 1));
            CHECK(hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, st));
            CHECK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, st));
            CHECK(hipStreamSynchronize(st));
            bad_h2d += count_diff((const uint32_t *)p, (const uint32_t *)pin, words) != 0;
            // ---- D2H into the same block: new device content first (from pinned memory)
            fill((uint32_t *)pin, words, (uint32_t)(2 * it + 2));
            memcpy(want, pin, bytes);
            CHECK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st));
            memset(p, 0, bytes);
            CHECK(hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, st));
            CHECK(hipStreamSynchronize(st));
            bad_d2h += count_diff((const uint32_t *)p, want, words) != 0;
            munmap(p, bytes);
        }
        printf("%9zu bytes: %d iterations, H2D mismatches %ld, D2H mismatches %ld\n", bytes, iters, bad_h2d, bad_d2h);
        bad_total += (bad_h2d || bad_d2h) ? 1 : 0;
        free(want);
        CHECK(hipFree(dev));
        CHECK(hipHostFree(pin));
    }
    // the heap variant: malloc / free of blocks above the mmap threshold, as numpy does
    {
        const size_t bytes = 3 << 20, words = bytes / 4;
        void *dev = nullptr, *pin = nullptr;
        CHECK(hipMalloc(&dev, bytes));
        CHECK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
        long bad = 0, same_addr = 0;
        void *last = nullptr;
        for (int it = 0; it < iters; ++it) {
            uint32_t *p = (uint32_t *)malloc(bytes);
            same_addr += p == last;
            last = p;
            fill(p, words, (uint32_t)(7 * it + 3));
            CHECK(hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, st));
            CHECK(hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, st));
            CHECK(hipStreamSynchronize(st));
            bad += count_diff(p, (const uint32_t *)pin, words) != 0;
            free(p);
        }
        printf("malloc/free %zu bytes: %d iterations (%ld at the previous address), mismatches %ld\n", bytes, iters, same_addr, bad);
        bad_total += bad ? 1 : 0;
        CHECK(hipFree(dev));
        CHECK(hipHostFree(pin));
    }
    printf(bad_total ? "PROBE: MISMATCH\n" : "PROBE: clean\n");
    return bad_total ? 1 : 0;
}
