// Microbenchmark (diagnostic only): how many small random reads per second the memory system serves when every wave
// reads inside a window of its own — the access pattern of the match copies of config 4a (a 4-byte read anywhere in the
// frame's 128 KiB of output, thousands of frames in flight).  Varies the live footprint (waves x window) across the
// L2 / Infinity Cache / HBM sizes.   hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// one wave per workgroup; wave w reads `iters` x 64 words at pseudo-random offsets inside [w * window, (w + 1) * window)
template <int BYTES>
__global__ void k_gather(const uint8_t* base, uint64_t window, uint32_t nwin, uint32_t iters, uint32_t* sink) {
    const uint8_t* w = base + (uint64_t)(blockIdx.x % nwin) * window;
    uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const uint32_t mask = (uint32_t)window - 1u;
    for (uint32_t i = 0; i < iters; i += 4) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { x = x * 1664525u + 1013904223u; o[k] = ((x >> 7) & mask) & ~(uint32_t)(BYTES - 1); }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (BYTES == 4) acc += *(const uint32_t*)(w + o[k]);
            else { const uint4 v = *(const uint4*)(w + o[k]); acc += v.x ^ v.w; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const uint64_t total = 4ull << 30;
    uint8_t* buf; uint32_t* sink;
    CHECK(hipMalloc(&buf, total)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t iters = 2048, grid = 16384;                          /* every CU full whatever the footprint */
    printf("%10s %8s %12s %10s %12s %12s\n", "window", "windows", "footprint MB", "ms", "G reads/s", "GB/s @32B");
    for (uint64_t window : {131072ull, 2097152ull}) {
        for (uint32_t nwin : {16u, 64u, 256u, 1024u, 2048u, 4096u, 8192u, 16384u}) {
            if (window * nwin > total) continue;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(64), 0, 0, buf, window, nwin, iters, sink);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) {
                    const double reads = (double)grid * 64 * iters;
                    printf("%10llu %8u %12.1f %10.3f %12.2f %12.1f\n", (unsigned long long)window, nwin, window * nwin / 1048576.0, ms, reads / ms / 1e6, reads * 32 / ms / 1e6);
                }
            }
        }
    }
    return 0;
}
