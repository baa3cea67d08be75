// Diagnostic: which (XCC, SE, SH, CU) ids the waves of a full-chip launch report (s_getreg HW_REG_HW_ID / HW_REG_XCC_ID on gfx950).
//   hipcc --offload-arch=gfx950 -O3 census.hip -o census && ./census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(64) k(uint32_t* out) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, all 32 bits
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
    __builtin_amdgcn_s_sleep(127);
    for (int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(127);                   // stay resident until the whole grid is
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    const int grid = 256 * 16;
    uint32_t* d; CHECK(hipMalloc(&d, grid * 8));
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(grid * 2); CHECK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, int> per_cu; std::map<uint32_t, int> cu_ids, se_ids, sh_ids;
    for (int i = 0; i < grid; i++) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 15, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++; cu_ids[cu]++; se_ids[se]++; sh_ids[sh]++;
    }
    printf("distinct (xcc, se, sh, cu): %zu\n", per_cu.size());
    printf("cu_id histogram:"); for (auto& kv : cu_ids) printf(" %u:%d", kv.first, kv.second); printf("\n");
    printf("se_id histogram:"); for (auto& kv : se_ids) printf(" %u:%d", kv.first, kv.second); printf("\n");
    printf("sh_id histogram:"); for (auto& kv : sh_ids) printf(" %u:%d", kv.first, kv.second); printf("\n");
    int even = 0, odd = 0; for (auto& kv : per_cu) { if (kv.first & 1) odd++; else even++; }
    printf("CUs with even cu_id %d, odd %d\n", even, odd);
    std::map<uint32_t, int> per_xcc; for (auto& kv : per_cu) per_xcc[kv.first >> 12]++;
    printf("CUs per xcc:"); for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second); printf("\n");
    for (auto& kv : per_cu) if ((kv.first >> 12) == 0) printf("  xcc 0 se %u sh %u cu %2u: %d waves\n", (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15, kv.second);
    return 0;
}
