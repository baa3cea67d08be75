// Diagnostic microbenchmark: what one SIMD of gfx950 issues per clock for the integer instruction
// kinds the decode kernels are made of, at 1 / 2 / 4 / 8 waves per SIMD, with EVERY CU busy.
//   hipcc --offload-arch=gfx950 -O3 issue.hip -o issue && ./issue
// A workgroup is 256 threads = one wave per SIMD; W workgroups per CU are forced by the dynamic LDS size (160 KiB / W each), the
// grid is 256 x W workgroups (all resident).  Every wave runs `iters` times a straight-line block of 64 instructions of one kind
// (the LDS streams: 64 + 8 waits).
// Reported: ns per wave-instruction as one wave sees it, and wave-instructions per clock and SIMD summed over its waves (the
// clock measured in the kernel: s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define R8(X) X X X X X X X X
// one instruction on eight registers round robin: OP dst, <operands with the register as first source>
#define RR(OP, TAIL) OP " v100, v100" TAIL "\n " OP " v101, v101" TAIL "\n " OP " v102, v102" TAIL "\n " OP " v103, v103" TAIL "\n " OP " v104, v104" TAIL "\n " OP " v105, v105" TAIL "\n " OP " v106, v106" TAIL "\n " OP " v107, v107" TAIL "\n"
// ... with the constant first (VOP2 shifts: v_lshlrev_b32 dst, amount, src)
#define RC(OP, HEAD) OP " v100, " HEAD ", v100\n " OP " v101, " HEAD ", v101\n " OP " v102, " HEAD ", v102\n " OP " v103, " HEAD ", v103\n " OP " v104, " HEAD ", v104\n " OP " v105, " HEAD ", v105\n " OP " v106, " HEAD ", v106\n " OP " v107, " HEAD ", v107\n"
#define I_AND   RR("v_and_b32", ", %[Y]")
#define I_XOR   RR("v_xor_b32", ", %[Y]")
#define I_SUB   RR("v_sub_u32", ", %[Y]")
#define I_MIN   RR("v_min_u32", ", %[Y]")
#define I_MOV   "v_mov_b32 v100, %[Y]\n v_mov_b32 v101, %[Y]\n v_mov_b32 v102, %[Y]\n v_mov_b32 v103, %[Y]\n v_mov_b32 v104, %[Y]\n v_mov_b32 v105, %[Y]\n v_mov_b32 v106, %[Y]\n v_mov_b32 v107, %[Y]\n"
#define I_SHL   RC("v_lshlrev_b32", "1")
#define I_SHR   RC("v_lshrrev_b32", "1")
#define I_SHLV  RC("v_lshlrev_b32", "%[Y]")
#define I_FFBH  "v_ffbh_u32 v100, v100\n v_ffbh_u32 v101, v101\n v_ffbh_u32 v102, v102\n v_ffbh_u32 v103, v103\n v_ffbh_u32 v104, v104\n v_ffbh_u32 v105, v105\n v_ffbh_u32 v106, v106\n v_ffbh_u32 v107, v107\n"
#define I_CNDM  RR("v_cndmask_b32", ", %[Y], vcc")
#define I_ADD64 RR("v_add_u32_e64", ", %[Y]")
#define I_ADDK  RC("v_add_u32", "0x12345")
#define I_ADD3  RR("v_add3_u32", ", %[Y], %[Z]")
#define I_LADD  RR("v_lshl_add_u32", ", 1, %[Y]")
#define I_ANDOR RR("v_and_or_b32", ", %[Y], %[Z]")
#define I_ALIGN RR("v_alignbit_b32", ", %[Y], 5")
#define I_BFI   RR("v_bfi_b32", ", %[Y], %[Z]")
#define I_MAD24 RR("v_mad_u32_u24", ", %[Y], %[Z]")
#define I_MUL24 RR("v_mul_u32_u24", ", %[Y]")
#define I_MULLO RR("v_mul_lo_u32", ", %[Y]")
#define I_SDWA  "v_add_u32_sdwa v100, v100, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa v101, v101, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa v102, v102, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa v103, v103, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa v104, v104, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa v105, v105, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa v106, v106, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_add_u32_sdwa v107, v107, %[Y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define I_QPERM "v_add_u32_dpp v100, v108, v100 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v101, v108, v101 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v102, v108, v102 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v103, v108, v103 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v104, v108, v104 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v105, v108, v105 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v106, v108, v106 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v107, v108, v107 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
#define I_CMPS  "v_cmp_lt_u32 vcc, v100, %[Y]\n v_cmp_lt_u32 vcc, v101, %[Y]\n v_cmp_lt_u32 vcc, v102, %[Y]\n v_cmp_lt_u32 vcc, v103, %[Y]\n v_cmp_lt_u32 vcc, v104, %[Y]\n v_cmp_lt_u32 vcc, v105, %[Y]\n v_cmp_lt_u32 vcc, v106, %[Y]\n v_cmp_lt_u32 vcc, v107, %[Y]\n"
#define I_BCNT  RR("v_bcnt_u32_b32", ", %[Y]")
#define I_MBCNT RR("v_mbcnt_lo_u32_b32", ", %[Y]")
#define I_SAND64 "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[26:27]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[26:27]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n"
#define I_SBCNT "s_bcnt1_i32_b64 s20, s[22:23]\n s_bcnt1_i32_b64 s21, s[22:23]\n s_bcnt1_i32_b64 s24, s[26:27]\n s_bcnt1_i32_b64 s25, s[26:27]\n s_bcnt1_i32_b64 s20, s[22:23]\n s_bcnt1_i32_b64 s21, s[22:23]\n s_bcnt1_i32_b64 s24, s[26:27]\n s_bcnt1_i32_b64 s25, s[26:27]\n"
#define I_LDSW  "ds_write_b32 v109, v100\n ds_write_b32 v109, v101 offset:256\n ds_write_b32 v109, v102 offset:512\n ds_write_b32 v109, v103 offset:768\n ds_write_b32 v109, v104 offset:1024\n ds_write_b32 v109, v105 offset:1280\n ds_write_b32 v109, v106 offset:1536\n ds_write_b32 v109, v107 offset:1792\n s_waitcnt lgkmcnt(0)\n"
#define I_LDSR64 "ds_read_b64 v[100:101], v110\n ds_read_b64 v[102:103], v110 offset:512\n ds_read_b64 v[104:105], v110 offset:1024\n ds_read_b64 v[106:107], v110 offset:1536\n ds_read_b64 v[100:101], v110 offset:2048\n ds_read_b64 v[102:103], v110 offset:2560\n ds_read_b64 v[104:105], v110 offset:3072\n ds_read_b64 v[106:107], v110 offset:3584\n s_waitcnt lgkmcnt(0)\n"
#define I_OR    RR("v_or_b32", ", %[Y]")
#define I_MAX   RR("v_max_u32", ", %[Y]")
#define I_ADDS  RC("v_add_u32", "s22")
#define I_ANDS  RC("v_and_b32", "s22")
#define I_BFES  "v_bfe_u32 v100, v100, s22, 5\n v_bfe_u32 v101, v101, s22, 5\n v_bfe_u32 v102, v102, s22, 5\n v_bfe_u32 v103, v103, s22, 5\n v_bfe_u32 v104, v104, s22, 5\n v_bfe_u32 v105, v105, s22, 5\n v_bfe_u32 v106, v106, s22, 5\n v_bfe_u32 v107, v107, s22, 5\n"
#define I_CNDS  RR("v_cndmask_b32_e64", ", %[Y], s[26:27]")
#define I_CNDV4 "v_cmp_lt_u32 vcc, v100, %[Y]\n v_add_u32 v104, v104, %[Y]\n v_add_u32 v105, v105, %[Y]\n v_add_u32 v106, v106, %[Y]\n v_cndmask_b32 v101, v101, %[Y], vcc\n v_add_u32 v107, v107, %[Y]\n v_add_u32 v102, v102, %[Y]\n v_add_u32 v103, v103, %[Y]\n"
#define I_CND2  "v_cmp_lt_u32 vcc, v100, %[Y]\n v_cndmask_b32 v101, v101, %[Y], vcc\n v_cndmask_b32 v102, v102, %[Y], vcc\n v_cndmask_b32 v103, v103, %[Y], vcc\n v_cmp_lt_u32 vcc, v104, %[Y]\n v_cndmask_b32 v105, v105, %[Y], vcc\n v_cndmask_b32 v106, v106, %[Y], vcc\n v_cndmask_b32 v107, v107, %[Y], vcc\n"
#define I_CNDSM "s_mov_b64 vcc, s[26:27]\n v_cndmask_b32 v101, v101, %[Y], vcc\n s_mov_b64 vcc, s[22:23]\n v_cndmask_b32 v103, v103, %[Y], vcc\n s_mov_b64 vcc, s[26:27]\n v_cndmask_b32 v105, v105, %[Y], vcc\n s_mov_b64 vcc, s[22:23]\n v_cndmask_b32 v107, v107, %[Y], vcc\n"
#define I_CMP64 "v_cmp_lt_u32_e64 s[20:21], v100, %[Y]\n v_cmp_lt_u32_e64 s[24:25], v101, %[Y]\n v_cmp_lt_u32_e64 s[20:21], v102, %[Y]\n v_cmp_lt_u32_e64 s[24:25], v103, %[Y]\n v_cmp_lt_u32_e64 s[20:21], v104, %[Y]\n v_cmp_lt_u32_e64 s[24:25], v105, %[Y]\n v_cmp_lt_u32_e64 s[20:21], v106, %[Y]\n v_cmp_lt_u32_e64 s[24:25], v107, %[Y]\n"
#define I_RFL   "v_readfirstlane_b32 s20, v100\n v_readfirstlane_b32 s21, v101\n v_readfirstlane_b32 s22, v102\n v_readfirstlane_b32 s23, v103\n v_readfirstlane_b32 s24, v104\n v_readfirstlane_b32 s25, v105\n v_readfirstlane_b32 s20, v106\n v_readfirstlane_b32 s21, v107\n"
#define I_ADDCO "v_add_co_u32 v100, vcc, v100, %[Y]\n v_add_co_u32 v101, vcc, v101, %[Y]\n v_add_co_u32 v102, vcc, v102, %[Y]\n v_add_co_u32 v103, vcc, v103, %[Y]\n v_add_co_u32 v104, vcc, v104, %[Y]\n v_add_co_u32 v105, vcc, v105, %[Y]\n v_add_co_u32 v106, vcc, v106, %[Y]\n v_add_co_u32 v107, vcc, v107, %[Y]\n"
#define I_ADD64B "v_add_co_u32 v100, vcc, v100, %[Y]\n v_addc_co_u32 v101, vcc, v101, %[Z], vcc\n v_add_co_u32 v102, vcc, v102, %[Y]\n v_addc_co_u32 v103, vcc, v103, %[Z], vcc\n v_add_co_u32 v104, vcc, v104, %[Y]\n v_addc_co_u32 v105, vcc, v105, %[Z], vcc\n v_add_co_u32 v106, vcc, v106, %[Y]\n v_addc_co_u32 v107, vcc, v107, %[Z], vcc\n"
#define I_LSHL64 "v_lshlrev_b64 v[100:101], 3, v[100:101]\n v_lshlrev_b64 v[102:103], 3, v[102:103]\n v_lshlrev_b64 v[104:105], 3, v[104:105]\n v_lshlrev_b64 v[106:107], 3, v[106:107]\n v_lshlrev_b64 v[100:101], 3, v[100:101]\n v_lshlrev_b64 v[102:103], 3, v[102:103]\n v_lshlrev_b64 v[104:105], 3, v[104:105]\n v_lshlrev_b64 v[106:107], 3, v[106:107]\n"
#define I_BPERM "ds_bpermute_b32 v100, v109, v100\n ds_bpermute_b32 v101, v109, v101\n ds_bpermute_b32 v102, v109, v102\n ds_bpermute_b32 v103, v109, v103\n ds_bpermute_b32 v104, v109, v104\n ds_bpermute_b32 v105, v109, v105\n ds_bpermute_b32 v106, v109, v106\n ds_bpermute_b32 v107, v109, v107\n s_waitcnt lgkmcnt(0)\n"
#define I_EXEC  "s_and_saveexec_b64 s[20:21], s[26:27]\n v_add_u32 v100, v100, %[Y]\n s_mov_b64 exec, s[20:21]\n v_add_u32 v101, v101, %[Y]\n s_and_saveexec_b64 s[20:21], s[26:27]\n v_add_u32 v102, v102, %[Y]\n s_mov_b64 exec, s[20:21]\n v_add_u32 v103, v103, %[Y]\n"
#define I_MIX44 "v_bfe_u32 v100, v100, 1, 31\n s_add_u32 s20, s20, 3\n v_bfe_u32 v101, v101, 1, 31\n s_add_u32 s21, s21, 3\n v_bfe_u32 v102, v102, 1, 31\n s_add_u32 s22, s22, 3\n v_bfe_u32 v103, v103, 1, 31\n s_add_u32 s23, s23, 3\n"
#define I_MIXFS "v_bfe_u32 v100, v100, 1, 31\n v_add_u32 v104, v104, %[Y]\n v_bfe_u32 v101, v101, 1, 31\n v_add_u32 v105, v105, %[Y]\n v_bfe_u32 v102, v102, 1, 31\n v_add_u32 v106, v106, %[Y]\n v_bfe_u32 v103, v103, 1, 31\n v_add_u32 v107, v107, %[Y]\n"
// independent: eight registers round robin; dependent: one register
#define I_ADD  "v_add_u32 v100, v100, %[Y]\n v_add_u32 v101, v101, %[Y]\n v_add_u32 v102, v102, %[Y]\n v_add_u32 v103, v103, %[Y]\n v_add_u32 v104, v104, %[Y]\n v_add_u32 v105, v105, %[Y]\n v_add_u32 v106, v106, %[Y]\n v_add_u32 v107, v107, %[Y]\n"
#define D_ADD  "v_add_u32 v100, v100, %[Y]\n v_add_u32 v100, v100, %[Y]\n v_add_u32 v100, v100, %[Y]\n v_add_u32 v100, v100, %[Y]\n v_add_u32 v100, v100, %[Y]\n v_add_u32 v100, v100, %[Y]\n v_add_u32 v100, v100, %[Y]\n v_add_u32 v100, v100, %[Y]\n"
#define I_BFE  "v_bfe_u32 v100, v100, 1, 31\n v_bfe_u32 v101, v101, 1, 31\n v_bfe_u32 v102, v102, 1, 31\n v_bfe_u32 v103, v103, 1, 31\n v_bfe_u32 v104, v104, 1, 31\n v_bfe_u32 v105, v105, 1, 31\n v_bfe_u32 v106, v106, 1, 31\n v_bfe_u32 v107, v107, 1, 31\n"
#define I_LOR  "v_lshl_or_b32 v100, v100, 1, %[Y]\n v_lshl_or_b32 v101, v101, 1, %[Y]\n v_lshl_or_b32 v102, v102, 1, %[Y]\n v_lshl_or_b32 v103, v103, 1, %[Y]\n v_lshl_or_b32 v104, v104, 1, %[Y]\n v_lshl_or_b32 v105, v105, 1, %[Y]\n v_lshl_or_b32 v106, v106, 1, %[Y]\n v_lshl_or_b32 v107, v107, 1, %[Y]\n"
#define I_PERM "v_perm_b32 v100, v100, %[Y], %[Z]\n v_perm_b32 v101, v101, %[Y], %[Z]\n v_perm_b32 v102, v102, %[Y], %[Z]\n v_perm_b32 v103, v103, %[Y], %[Z]\n v_perm_b32 v104, v104, %[Y], %[Z]\n v_perm_b32 v105, v105, %[Y], %[Z]\n v_perm_b32 v106, v106, %[Y], %[Z]\n v_perm_b32 v107, v107, %[Y], %[Z]\n"
#define I_DPP  "v_add_u32_dpp v100, v108, v100 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v101, v108, v101 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v102, v108, v102 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v103, v108, v103 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v104, v108, v104 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v105, v108, v105 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v106, v108, v106 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v107, v108, v107 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CMP  "v_cmp_lt_u32 vcc, v100, %[Y]\n v_cndmask_b32 v101, v101, %[Y], vcc\n v_cmp_lt_u32 vcc, v102, %[Y]\n v_cndmask_b32 v103, v103, %[Y], vcc\n v_cmp_lt_u32 vcc, v104, %[Y]\n v_cndmask_b32 v105, v105, %[Y], vcc\n v_cmp_lt_u32 vcc, v106, %[Y]\n v_cndmask_b32 v107, v107, %[Y], vcc\n"
#define I_SALU "s_add_u32 s20, s20, 3\n s_add_u32 s21, s21, 3\n s_add_u32 s22, s22, 3\n s_add_u32 s23, s23, 3\n s_add_u32 s24, s24, 3\n s_add_u32 s25, s25, 3\n s_add_u32 s26, s26, 3\n s_add_u32 s27, s27, 3\n"
#define D_SALU "s_add_u32 s20, s20, 3\n s_add_u32 s20, s20, 3\n s_add_u32 s20, s20, 3\n s_add_u32 s20, s20, 3\n s_add_u32 s20, s20, 3\n s_add_u32 s20, s20, 3\n s_add_u32 s20, s20, 3\n s_add_u32 s20, s20, 3\n"
#define I_MIX  "v_add_u32 v100, v100, %[Y]\n s_add_u32 s20, s20, 3\n v_add_u32 v101, v101, %[Y]\n s_add_u32 s21, s21, 3\n v_add_u32 v102, v102, %[Y]\n s_add_u32 s22, s22, 3\n v_add_u32 v103, v103, %[Y]\n s_add_u32 s23, s23, 3\n"
#define I_MIX31 "v_add_u32 v100, v100, %[Y]\n v_add_u32 v101, v101, %[Y]\n v_add_u32 v102, v102, %[Y]\n s_add_u32 s20, s20, 3\n v_add_u32 v103, v103, %[Y]\n v_add_u32 v104, v104, %[Y]\n v_add_u32 v105, v105, %[Y]\n s_add_u32 s21, s21, 3\n"
#define I_RDL  "v_readlane_b32 s20, v100, 3\n v_readlane_b32 s21, v101, 3\n v_readlane_b32 s22, v102, 3\n v_readlane_b32 s23, v103, 3\n v_readlane_b32 s24, v104, 3\n v_readlane_b32 s25, v105, 3\n v_readlane_b32 s26, v106, 3\n v_readlane_b32 s27, v107, 3\n"
#define I_LDSR "ds_read_b32 v100, v109\n ds_read_b32 v101, v109 offset:256\n ds_read_b32 v102, v109 offset:512\n ds_read_b32 v103, v109 offset:768\n ds_read_b32 v104, v109 offset:1024\n ds_read_b32 v105, v109 offset:1280\n ds_read_b32 v106, v109 offset:1536\n ds_read_b32 v107, v109 offset:1792\n s_waitcnt lgkmcnt(0)\n"

#define KERNEL(NAME, BODY) \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t iters, unsigned long long* clk) { \
    extern __shared__ uint32_t lds[]; \
    uint32_t x = threadIdx.x, y = out[0] | 1u, z = 0x02010003u, acc = 0; \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); \
    asm volatile("v_mov_b32 v100, %[X]\n v_mov_b32 v101, %[X]\n v_mov_b32 v102, %[X]\n v_mov_b32 v103, %[X]\n v_mov_b32 v104, %[X]\n v_mov_b32 v105, %[X]\n v_mov_b32 v106, %[X]\n v_mov_b32 v107, %[X]\n v_mov_b32 v108, %[Y]\n v_lshlrev_b32 v109, 2, %[X]\n v_lshlrev_b32 v110, 3, %[X]\n" \
                 "s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n s_mov_b32 s22, 0\n s_mov_b32 s23, 0\n s_mov_b32 s24, 0\n s_mov_b32 s25, 0\n s_mov_b32 s26, 0\n s_mov_b32 s27, 0\n" \
                 "s_mov_b32 s28, %[N]\n s_mov_b32 s26, 0x0f0f0f0f\n s_mov_b32 s27, -1\n s_mov_b32 s22, 5\n s_mov_b32 s23, 0x00ff00ff\n" \
                 "1:\n" R8(BODY) \
                 "s_sub_u32 s28, s28, 1\n s_cmp_lg_u32 s28, 0\n s_cbranch_scc1 1b\n" \
                 "v_add_u32 %[A], v100, v101\n v_add_u32 %[A], %[A], v102\n v_add_u32 %[A], %[A], v103\n v_add_u32 %[A], %[A], v104\n v_add_u32 %[A], %[A], v105\n v_add_u32 %[A], %[A], v106\n v_add_u32 %[A], %[A], v107\n v_add_u32 %[A], %[A], s20\n v_add_u32 %[A], %[A], s21\n" \
                 : [A] "=v"(acc) : [X] "v"(x), [Y] "v"(y), [Z] "v"(z), [N] "s"(iters) \
                 : "memory", "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", \
                   "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28"); \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); \
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; } \
    if (acc == 0x12345u) lds[threadIdx.x] = acc; \
    out[1 + threadIdx.x + blockIdx.x * blockDim.x] = acc; \
}
KERNEL(k_add_i, I_ADD)   KERNEL(k_add_d, D_ADD)   KERNEL(k_bfe, I_BFE)     KERNEL(k_lor, I_LOR)   KERNEL(k_perm, I_PERM)
KERNEL(k_dpp, I_DPP)     KERNEL(k_cmp, I_CMP)     KERNEL(k_salu_i, I_SALU) KERNEL(k_salu_d, D_SALU) KERNEL(k_mix, I_MIX)
KERNEL(k_mix31, I_MIX31) KERNEL(k_rdl, I_RDL)     KERNEL(k_ldsr, I_LDSR)
KERNEL(k_and, I_AND) KERNEL(k_xor, I_XOR) KERNEL(k_sub, I_SUB) KERNEL(k_min, I_MIN) KERNEL(k_mov, I_MOV) KERNEL(k_shl, I_SHL) KERNEL(k_shr, I_SHR) KERNEL(k_shlv, I_SHLV)
KERNEL(k_ffbh, I_FFBH) KERNEL(k_cndm, I_CNDM) KERNEL(k_add64, I_ADD64) KERNEL(k_addk, I_ADDK) KERNEL(k_add3, I_ADD3) KERNEL(k_ladd, I_LADD) KERNEL(k_andor, I_ANDOR)
KERNEL(k_align, I_ALIGN) KERNEL(k_bfi, I_BFI) KERNEL(k_mad24, I_MAD24) KERNEL(k_mul24, I_MUL24) KERNEL(k_mullo, I_MULLO) KERNEL(k_sdwa, I_SDWA) KERNEL(k_qperm, I_QPERM)
KERNEL(k_or, I_OR) KERNEL(k_max, I_MAX) KERNEL(k_adds, I_ADDS) KERNEL(k_ands, I_ANDS) KERNEL(k_bfes, I_BFES) KERNEL(k_cnds, I_CNDS) KERNEL(k_cndv4, I_CNDV4) KERNEL(k_cnd2, I_CND2)
KERNEL(k_cndsm, I_CNDSM) KERNEL(k_cmp64, I_CMP64) KERNEL(k_rfl, I_RFL) KERNEL(k_addco, I_ADDCO) KERNEL(k_add64b, I_ADD64B) KERNEL(k_lshl64, I_LSHL64) KERNEL(k_bperm, I_BPERM) KERNEL(k_exec, I_EXEC)
KERNEL(k_mix44, I_MIX44) KERNEL(k_mixfs, I_MIXFS)
KERNEL(k_cmps, I_CMPS) KERNEL(k_bcnt, I_BCNT) KERNEL(k_mbcnt, I_MBCNT) KERNEL(k_sand64, I_SAND64) KERNEL(k_sbcnt, I_SBCNT) KERNEL(k_ldsw, I_LDSW) KERNEL(k_ldsr64, I_LDSR64)

typedef void (*kern_t)(uint32_t*, uint32_t, unsigned long long*);
struct Kind { const char* name; kern_t k; int per_block; };
int main(int argc, char** argv) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d; unsigned long long* clk; CHECK(hipMalloc(&d, 64 << 20)); CHECK(hipMemset(d, 0, 64 << 20)); CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const Kind kinds[] = {
        {"v_add_u32, 8 independent registers", k_add_i, 64}, {"v_add_u32, one dependent register", k_add_d, 64}, {"v_bfe_u32", k_bfe, 64},
        {"v_lshl_or_b32", k_lor, 64}, {"v_perm_b32", k_perm, 64}, {"v_add_u32_dpp row_shr:1", k_dpp, 64}, {"v_cmp + v_cndmask pairs", k_cmp, 64},
        {"s_add_u32, 8 independent registers", k_salu_i, 64}, {"s_add_u32, one dependent register", k_salu_d, 64},
        {"v_add_u32 / s_add_u32 alternating", k_mix, 64}, {"3 v_add_u32 : 1 s_add_u32", k_mix31, 64}, {"v_readlane_b32", k_rdl, 64},
        {"8 ds_read_b32 + s_waitcnt lgkmcnt(0)", k_ldsr, 72},
        {"v_and_b32 (VOP2)", k_and, 64}, {"v_xor_b32 (VOP2)", k_xor, 64}, {"v_sub_u32 (VOP2)", k_sub, 64}, {"v_min_u32 (VOP2)", k_min, 64}, {"v_mov_b32 (VOP1)", k_mov, 64},
        {"v_lshlrev_b32 by constant (VOP2)", k_shl, 64}, {"v_lshrrev_b32 by constant (VOP2)", k_shr, 64}, {"v_lshlrev_b32 by register (VOP2)", k_shlv, 64},
        {"v_ffbh_u32 (VOP1)", k_ffbh, 64}, {"v_cndmask_b32 (VOP2, vcc)", k_cndm, 64}, {"v_add_u32 in VOP3 encoding", k_add64, 64}, {"v_add_u32 with a 32-bit literal", k_addk, 64},
        {"v_add3_u32 (VOP3)", k_add3, 64}, {"v_lshl_add_u32 (VOP3)", k_ladd, 64}, {"v_and_or_b32 (VOP3)", k_andor, 64}, {"v_alignbit_b32 (VOP3)", k_align, 64}, {"v_bfi_b32 (VOP3)", k_bfi, 64},
        {"v_mad_u32_u24 (VOP3)", k_mad24, 64}, {"v_mul_u32_u24 (VOP2)", k_mul24, 64}, {"v_mul_lo_u32 (VOP3)", k_mullo, 64}, {"v_add_u32_sdwa", k_sdwa, 64},
        {"v_add_u32_dpp quad_perm", k_qperm, 64}, {"v_cmp_lt_u32 vcc (VOPC)", k_cmps, 64}, {"v_bcnt_u32_b32 (VOP3)", k_bcnt, 64}, {"v_mbcnt_lo_u32_b32 (VOP3)", k_mbcnt, 64},
        {"v_or_b32 (VOP2)", k_or, 64}, {"v_max_u32 (VOP2)", k_max, 64}, {"v_add_u32 with an SGPR operand", k_adds, 64}, {"v_and_b32 with an SGPR operand", k_ands, 64},
        {"v_bfe_u32 with an SGPR operand", k_bfes, 64}, {"v_cndmask_b32_e64, mask in an SGPR pair", k_cnds, 64}, {"v_cmp, 3 v_add, v_cndmask, 3 v_add", k_cndv4, 64},
        {"v_cmp + 3 v_cndmask on its vcc", k_cnd2, 64}, {"s_mov_b64 vcc + v_cndmask pairs", k_cndsm, 64}, {"v_cmp_lt_u32_e64 -> SGPR pair", k_cmp64, 64},
        {"v_readfirstlane_b32", k_rfl, 64}, {"v_add_co_u32 (carry out to vcc)", k_addco, 64}, {"v_add_co_u32 + v_addc_co_u32 (64-bit add)", k_add64b, 64},
        {"v_lshlrev_b64", k_lshl64, 64}, {"8 ds_bpermute_b32 + s_waitcnt", k_bperm, 72}, {"s_and_saveexec / v_add / s_mov exec / v_add", k_exec, 64},
        {"v_bfe_u32 / s_add_u32 alternating", k_mix44, 64}, {"v_bfe_u32 / v_add_u32 alternating", k_mixfs, 64},
        {"s_and_b64", k_sand64, 64}, {"s_bcnt1_i32_b64", k_sbcnt, 64}, {"8 ds_write_b32 + s_waitcnt lgkmcnt(0)", k_ldsw, 72}, {"8 ds_read_b64 + s_waitcnt lgkmcnt(0)", k_ldsr64, 72}};
    const uint32_t iters = 20000;
    printf("%d CUs; 64 instructions x %u iterations per wave; every CU holds W workgroups of 4 waves (one per SIMD)\n", cus, iters);
    printf("%-40s %5s %12s %14s %16s %14s %10s\n", "instruction stream", "W", "kernel ms", "ns/instr/wave", "instr/clk/SIMD", "clk/instr/SIMD", "clock GHz");
    for (const Kind& kd : kinds) {
        CHECK(hipFuncSetAttribute((const void*)kd.k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int W : {1, 2, 4, 8}) {
            const size_t lds = (size_t)(160 * 1024 / W) & ~(size_t)255;
            const int grid = cus * W;
            hipLaunchKernelGGL(kd.k, dim3(grid), dim3(256), lds, 0, d, 10u, clk);
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(kd.k, dim3(grid), dim3(256), lds, 0, d, iters, clk); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;      /* s_memrealtime ticks at 100 MHz */
            const double n = (double)iters * kd.per_block;             /* instructions per wave: the body is 8 (or 9) instructions, repeated 8 times per iteration */
            const double ns_wave = ms * 1e6 / n, per_clk = (double)W * n / (ms * 1e6 * ghz);
            printf("%-40s %5d %12.3f %14.3f %16.3f %14.2f %10.2f\n", kd.name, W, ms, ns_wave, per_clk, 1.0 / per_clk, ghz);
        }
    }
    return 0;
}
