// Diagnostic microbenchmark: cost of the pieces of the chain step (czstd_chain.hip, czc_group_asm) on ONE wave
// per SIMD.  Variants remove / replace pieces of the step; the chain data is synthetic (a table whose
// entries keep every state in range), so only time is meaningful.
//   hipcc --offload-arch=gfx950 -O3 chain_step.hip -o chain_step && ./chain_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// pieces of the current step (czstd_chain.hip, CZC_ASM_HEAD / CZC_ASM_TAIL)
#define H1 "s_waitcnt lgkmcnt(3)\n" "v_ffbh_u32 v100, %[E]\n" "v_and_or_b32 v101, %[E], %[XM], %[K64]\n" "v_lshrrev_b32 v103, 22, %[E]\n" "v_sub_u32 v102, v101, v100\n"
#define NOP1 "s_nop 1\n"
#define NOP0 "s_nop 0\n"
#define H_DPP \
    "v_add_u32_dpp v105, v102, v102 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v106, v102, %[M1] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v107, v102, %[M2] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp v105, v102, v105 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n"
#define H2 "v_add3_u32 v109, v106, v107, v102\n" "v_bfe_u32 v108, v105, 16, 8\n" "s_waitcnt lgkmcnt(0)\n" "v_cmp_ge_u32 vcc, %[PH], v108\n" "v_sub_u32 v110, %[PH], v108\n"
#define H3 "v_cndmask_b32 v111, %[W1], %[W2], vcc\n" "v_cndmask_b32 v112, %[W0], %[W1], vcc\n" "v_alignbit_b32 v113, v111, v112, v110\n" \
    "v_bfe_u32 v116, v113, v109, v100\n" "v_lshl_or_b32 %[S], v103, v100, v116\n" "v_lshl_add_u32 v117, %[S], 1, %[TB]\n" "ds_read_u16_d16_hi %[E], v117\n"
#define T_WIN "v_alignbit_b32 v120, %[W2], %[W1], %[PH]\n"
#define T_MAX "v_max_u32 %[SLOW], %[SLOW], v108\n"
#define T_DOT "v_dot4c_i32_i8_e32 %[U], 0x01ff0001, v105\n"
#define T_C "v_lshl_add_u32 v118, %[S], %[SH], %[NK]\n"
#define T_STORE "global_store_dwordx2 %[RP], v[120:121], off\n"
#define T_DPP "v_add_u32_dpp v121, v118, v118 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" "v_add_u32_dpp v121, v118, v121 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n"
#define T_ADDR "v_bfe_u32 v114, %[U], 5, 7\n" "v_and_b32 %[PH], 31, %[U]\n" "v_lshl_add_u32 v115, v114, 2, %[RB]\n"
#define T_RING "ds_read_b32 %[W0], v115\n" "ds_read_b32 %[W1], v115 offset:4\n" "ds_read_b32 %[W2], v115 offset:8\n"
#define T_RING1 "ds_read_b32 %[W0], v115\n" "ds_read_b32 %[W1], v115 offset:4\n" "s_nop 0\n"
#define T_RING0 "s_nop 0\n" "s_nop 0\n" "s_nop 0\n"
#define HEAD_FULL H1 NOP1 H_DPP H2 NOP0 H3
#ifndef VARIANT
#define VARIANT 0
#endif
#if VARIANT == 0   // current step
#define STEP HEAD_FULL T_WIN T_MAX T_DOT T_C T_STORE NOP0 T_DPP T_ADDR T_RING
#elif VARIANT == 1 // no store
#define STEP HEAD_FULL T_WIN T_MAX T_DOT T_C NOP0 T_DPP T_ADDR T_RING
#elif VARIANT == 2 // no store, no ring reads (3 nops instead)
#define STEP HEAD_FULL T_WIN T_MAX T_DOT T_C NOP0 T_DPP T_ADDR T_RING0
#elif VARIANT == 3 // no store, no ring reads, no nops at all
#define STEP H1 H_DPP H2 H3 T_WIN T_MAX T_DOT T_C T_DPP T_ADDR
#elif VARIANT == 4 // variant 3 without the 6 DPP ops
#define STEP H1 H2 H3 T_WIN T_MAX T_DOT T_C T_ADDR
#elif VARIANT == 5 // variant 3 without dot4c
#define STEP H1 H_DPP H2 H3 T_WIN T_MAX T_C T_DPP T_ADDR
#elif VARIANT == 6 // only the table chase: ffbh, shift, or, address, load
#define STEP "s_waitcnt lgkmcnt(0)\n" "v_ffbh_u32 v100, %[E]\n" "v_lshrrev_b32 v103, 22, %[E]\n" "v_lshl_or_b32 %[S], v103, v100, v100\n" "v_and_b32 %[S], 0x3ff, %[S]\n" "v_lshl_add_u32 v117, %[S], 1, %[TB]\n" "ds_read_u16_d16_hi %[E], v117\n"
#elif VARIANT == 7 // table chase + 20 plain VALU in the shadow
#define V5 "v_add_u32 v110, v110, v108\n" "v_add_u32 v111, v111, v110\n" "v_add_u32 v112, v112, v111\n" "v_add_u32 v113, v113, v112\n" "v_add_u32 v108, v108, v113\n"
#define STEP "s_waitcnt lgkmcnt(0)\n" "v_ffbh_u32 v100, %[E]\n" "v_lshrrev_b32 v103, 22, %[E]\n" "v_lshl_or_b32 %[S], v103, v100, v100\n" "v_and_b32 %[S], 0x3ff, %[S]\n" "v_lshl_add_u32 v117, %[S], 1, %[TB]\n" "ds_read_u16_d16_hi %[E], v117\n" V5 V5 V5 V5
#elif VARIANT == 8 // table chase + 40 plain VALU
#define V5 "v_add_u32 v110, v110, v108\n" "v_add_u32 v111, v111, v110\n" "v_add_u32 v112, v112, v111\n" "v_add_u32 v113, v113, v112\n" "v_add_u32 v108, v108, v113\n"
#define STEP "s_waitcnt lgkmcnt(0)\n" "v_ffbh_u32 v100, %[E]\n" "v_lshrrev_b32 v103, 22, %[E]\n" "v_lshl_or_b32 %[S], v103, v100, v100\n" "v_and_b32 %[S], 0x3ff, %[S]\n" "v_lshl_add_u32 v117, %[S], 1, %[TB]\n" "ds_read_u16_d16_hi %[E], v117\n" V5 V5 V5 V5 V5 V5 V5 V5
#endif
#if VARIANT >= 9
/* round 5: the step with the field picked by two 64-bit shifts (no compare, no selects, no phase register) */
#define N1 "s_waitcnt lgkmcnt(3)\n" "v_ffbh_u32 v100, %[E]\n" "v_bfe_u32 v101, %[E], 16, 5\n" "v_lshl_or_b32 v102, v100, 8, v101\n" "v_lshrrev_b32 v131, 22, %[E]\n"
#define N_DPP \
    "v_add_u32_dpp v105, v102, v102 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v106, v100, %[M1] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v107, v100, %[M2] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp v105, v102, v105 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n"
#define N2 "v_add3_u32 v109, v106, v107, v105\n" "s_waitcnt lgkmcnt(0)\n" \
    "v_alignbit_b32 v120, %[W2], %[W1], %[U]\n" "v_alignbit_b32 v128, %[W1], %[W0], %[U]\n" "v_mov_b32 v129, v120\n" \
    "v_lshlrev_b64 v[128:129], v109, v[128:129]\n" "v_mov_b32 v130, v129\n" "v_lshlrev_b64 v[132:133], v100, v[130:131]\n" \
    "v_mov_b32 %[S], v133\n" "v_lshl_add_u32 v117, v133, 1, %[TB]\n" "ds_read_u16_d16_hi %[E], v117\n"
#define N_WIN ""
#define N_DOT "v_dot4c_i32_i8_e32 %[U], 0xffff, v105\n"
#define N_MAX "v_max_u32_sdwa %[SLOW], %[SLOW], v105 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
#define N_ADDR "v_bfe_u32 v114, %[U], 5, 7\n" "v_lshl_add_u32 v115, v114, 2, %[RB]\n"
#undef STEP
#if VARIANT == 9    // new step, with the store (8 bytes per step, as variant 0) and the wide check
#define STEP N1 N_DPP N2 N_MAX N_DOT T_C T_STORE NOP0 T_DPP N_ADDR T_RING
#elif VARIANT == 10 // new step without the store
#define STEP N1 N_DPP N2 N_MAX N_DOT T_C NOP0 T_DPP N_ADDR T_RING
#elif VARIANT == 11 // ... and without the wide check
#define STEP N1 N_DPP N2 N_DOT T_C NOP0 T_DPP N_ADDR T_RING
#endif
#endif
#define STEP4 STEP STEP STEP STEP
#define STEP32 STEP4 STEP4 STEP4 STEP4 STEP4 STEP4 STEP4 STEP4

__global__ void __launch_bounds__(64, 1) k(uint32_t* out, uint64_t* sink, uint32_t groups) {
    __shared__ uint16_t tab[4096];
    __shared__ uint32_t ring[256];
    for (uint32_t i = threadIdx.x; i < 4096; i += 64) tab[i] = (uint16_t)(0x8000u >> (i % 5) | (i * 37 % 3) | ((i * 13) & 0x1C0));   // marker at 15..11, a few low v bits
    for (uint32_t i = threadIdx.x; i < 256; i += 64) ring[i] = i * 2654435761u;
    __syncthreads();
    uint32_t E = 0x80000000u, S = 512, U = 4096 * 8 + threadIdx.x, PH = U & 31, W0 = 1, W1 = 2, W2 = 3, SLOW = 0;
    const uint32_t tb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)tab;
    const uint32_t rb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)ring;
    uint64_t* rp = sink + (threadIdx.x >> 2) * 8 + blockIdx.x * 1024;
    const uint32_t r = threadIdx.x & 3;
    for (uint32_t g = 0; g < groups; g++) {
        asm volatile(STEP32 "s_waitcnt lgkmcnt(0)\n"
            : [E] "+v"(E), [S] "+v"(S), [U] "+v"(U), [PH] "+v"(PH), [W0] "+v"(W0), [W1] "+v"(W1), [W2] "+v"(W2), [SLOW] "+v"(SLOW)
            : [TB] "v"(tb - 1024u), [RB] "v"(rb), [M1] "v"(r ? 0xFFu : 0u), [M2] "v"(r == 2 ? 0xFFu : 0u), [SH] "v"(r * 9u), [NK] "v"(0u - (512u << (r * 9u))),
              [K64] "v"(r == 0 ? 0x40000040u : 0x40u), [XM] "s"(0x1F0000u), [RP] "v"(rp)
            : "memory", "vcc", "v100", "v101", "v102", "v103", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
              "v116", "v117", "v118", "v120", "v121", "v122", "v123", "v127", "v128", "v129", "v130", "v131", "v132", "v133");
        S = (S & 1023u) | 512u; U |= 0x8000u;                          // keep addresses in range
    }
    out[threadIdx.x + blockIdx.x * 64] = E + S + U + PH + W0 + SLOW;
}
int main() {
    uint32_t* d; uint64_t* sink; CHECK(hipMalloc(&d, 1 << 20)); CHECK(hipMalloc(&sink, 64 << 20));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const uint32_t groups = 2000; float ms;
    for (int grid : {1, 1024}) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, sink, 10u);
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, sink, groups); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d grid %4d: %.2f ns per step\n", VARIANT, grid, ms * 1e6 / (groups * 32.0));
    }
    return 0;
}
