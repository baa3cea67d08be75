// Microbenchmarks for the chain kernels' cost model (diagnostic only): single-wave dependent VALU
// chains, LDS pointer chase, and both with partial exec masks.  hipcc --offload-arch=gfx950 -O3 micro.hip -o micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_valu_dep(uint32_t* out, uint32_t iters, uint32_t lanes) {
    uint32_t x = threadIdx.x + 1, y = out[0];
    if (threadIdx.x < lanes) for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = __builtin_amdgcn_ubfe(x ^ y, 1, 31) + 3;   // 3 dependent VALU ops
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
__global__ void k_lds_chase(uint32_t* out, uint32_t iters, uint32_t lanes, uint32_t stride) {
    __shared__ uint32_t tab[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = (i * stride + 17) & 8191;
    __syncthreads();
    uint32_t x = threadIdx.x * 97 & 8191;
    if (threadIdx.x < lanes) for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = tab[x];
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
__global__ void k_lds_chase16(uint32_t* out, uint32_t iters, uint32_t lanes, uint32_t stride) {
    __shared__ uint16_t tab[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = (uint16_t)((i * stride + 17) & 8191);
    __syncthreads();
    uint32_t x = threadIdx.x * 97 & 8191;
    if (threadIdx.x < lanes) for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++) x = tab[x];
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
// independent VALU work (ILP 4) to see the issue rate of one wave
__global__ void k_valu_ilp(uint32_t* out, uint32_t iters, uint32_t lanes) {
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, y = out[0];
    if (threadIdx.x < lanes) for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { a = (a ^ y) + 3; b = (b ^ y) + 5; c = (c ^ y) + 7; d = (d ^ y) + 9; }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = a + b + c + d;
}
int main() {
    uint32_t* d; CHECK(hipMalloc(&d, 1 << 20)); CHECK(hipMemset(d, 0, 1 << 20));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int clk = 0; CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("clock rate attribute: %d kHz\n", clk);
    const uint32_t iters = 20000;
    float ms;
    for (int grid : {1, 256, 1024}) for (uint32_t lanes : {64u, 16u, 1u}) {
        hipLaunchKernelGGL(k_valu_dep, dim3(grid), dim3(64), 0, 0, d, 10u, lanes);
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_valu_dep, dim3(grid), dim3(64), 0, 0, d, iters, lanes); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("valu_dep   grid %4d lanes %2u: %.3f ns per dependent VALU op (96 per iter)\n", grid, lanes, ms * 1e6 / (iters * 96.0));
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_valu_ilp, dim3(grid), dim3(64), 0, 0, d, iters, lanes); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("valu_ilp4  grid %4d lanes %2u: %.3f ns per VALU op (64 per iter)\n", grid, lanes, ms * 1e6 / (iters * 64.0));
        for (uint32_t stride : {1u, 33u}) {
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_chase, dim3(grid), dim3(64), 0, 0, d, iters, lanes, stride); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("lds_chase  grid %4d lanes %2u stride %2u: %.2f ns per hop\n", grid, lanes, stride, ms * 1e6 / (iters * 32.0));
        }
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_chase16, dim3(grid), dim3(64), 0, 0, d, iters, lanes, 33u); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("lds_chase16 grid %4d lanes %2u: %.2f ns per hop\n", grid, lanes, ms * 1e6 / (iters * 32.0));
    }
    return 0;
}
