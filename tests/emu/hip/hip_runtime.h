/*
 * TEST INFRASTRUCTURE ONLY — a minimal CPU stand-in for the handful of HIP device
 * constructs czstd_kernels.hip uses, so the *unmodified kernel source* can be compiled with
 * g++ and run under AddressSanitizer/UBSan (GPU sanitizers are not available on the pool).
 * One emulated workgroup at a time, 64 pthreads per wave = 64 lanes, pthread barriers for
 * __syncthreads and the cross-lane intrinsics.  Never linked into the product library.
 */
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define CZ_DYNAMIC_LDS(name) static uint32_t name[40960]   /* stand-in for `extern __shared__` dynamic LDS (160 KiB) */
#define CZ_EMU 1
#include <sched.h>
#define __forceinline__ inline
#define __launch_bounds__(...)

struct emu_dim3 { unsigned x, y, z; };
extern thread_local emu_dim3 threadIdx;
extern thread_local emu_dim3 blockIdx;
extern emu_dim3 gridDim;                  /* blocks of the launch in flight (set by the harness) */
extern emu_dim3 blockDim;                 /* threads per block of the launch in flight */
struct uint4 { uint32_t x, y, z, w; };

/* A workgroup is EMU_MAX_WAVES waves at most: emu_barrier joins all its threads (__syncthreads), emu_wbar[w]
 * the 64 lanes of wave w (cross-lane intrinsics, wave barrier). */
#define EMU_MAX_WAVES 16
#define EMU_MAX_THREADS (64 * EMU_MAX_WAVES)
extern pthread_barrier_t emu_barrier;
extern pthread_barrier_t emu_wbar[EMU_MAX_WAVES];
extern volatile uint64_t emu_xchg_all[EMU_MAX_WAVES][64];
#define emu_xchg (emu_xchg_all[threadIdx.x >> 6])
#define EMU_LANE (threadIdx.x & 63u)

extern void* volatile emu_site[EMU_MAX_THREADS];          /* last barrier site per lane (hang diagnosis) */
extern volatile uint64_t emu_sync_count[EMU_MAX_THREADS];
extern void* volatile emu_ring[EMU_MAX_THREADS][64];
static inline void emu_note(void* site) { emu_ring[threadIdx.x][emu_sync_count[threadIdx.x] & 63] = site; }
static inline void emu_sync() { emu_sync_count[threadIdx.x]++; pthread_barrier_wait(&emu_wbar[threadIdx.x >> 6]); }
static inline void emu_sync_wg() { emu_sync_count[threadIdx.x]++; pthread_barrier_wait(&emu_barrier); }
#define __syncthreads() do { __label__ emu_here; emu_here: emu_site[threadIdx.x] = &&emu_here; emu_note(&&emu_here); emu_sync_wg(); } while (0)
template <class T> __attribute__((noinline)) static T __shfl(T v, int src) {
    emu_note(__builtin_return_address(0));
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T)); emu_xchg[EMU_LANE] = raw; emu_sync();
    uint64_t r = emu_xchg[src & 63]; emu_sync(); T out; memcpy(&out, &r, sizeof(T)); return out;
}
template <class T> __attribute__((noinline)) static T __shfl_up(T v, unsigned d) {
    emu_note(__builtin_return_address(0));
    uint64_t raw = 0; memcpy(&raw, &v, sizeof(T)); emu_xchg[EMU_LANE] = raw; emu_sync();
    uint64_t r = EMU_LANE >= d ? emu_xchg[EMU_LANE - d] : raw; emu_sync(); T out; memcpy(&out, &r, sizeof(T)); return out;
}
/* DPP (data-parallel primitives) as the kernels use them: quad_perm (0x00-0xFF), row_shr:n (0x110+n), wave_shr:1 (0x138),
 * row_bcast:15 (0x142), row_bcast:31 (0x143); rows are 16 lanes, banks 4 lanes.  A lane whose row or
 * bank is masked off, or whose source lane does not exist, keeps `old` (bound_ctrl:0 semantics). */
__attribute__((noinline)) static int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    emu_note(__builtin_return_address(0));
    emu_xchg[EMU_LANE] = (uint32_t)src; emu_sync();
    const int l = (int)EMU_LANE, row = l >> 4, pos = l & 15;
    int from = -1;
    if (ctrl >= 0 && ctrl <= 0xFF) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);   /* quad_perm:[a,b,c,d] */
    else if (ctrl > 0x110 && ctrl <= 0x11F) { const int n = ctrl - 0x110; from = pos - n >= 0 ? l - n : -1; }
    else if (ctrl == 0x138) from = l - 1;
    else if (ctrl == 0x142) from = row >= 1 ? row * 16 - 1 : -1;
    else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;
    else __builtin_trap();
    int out = old;
    if (((row_mask >> row) & 1) && ((bank_mask >> (pos >> 2)) & 1)) out = from >= 0 ? (int)(uint32_t)emu_xchg[from] : (bound_ctrl ? 0 : old);
    emu_sync(); return out;
}
__attribute__((noinline)) static int __builtin_amdgcn_readlane(int v, int lane) {
    emu_note(__builtin_return_address(0));
    emu_xchg[EMU_LANE] = (uint32_t)v; emu_sync();
    const int r = (int)(uint32_t)emu_xchg[lane & 63]; emu_sync(); return r;
}
__attribute__((noinline)) static unsigned long long __ballot(int pred) {
    emu_note(__builtin_return_address(0));
    emu_xchg[EMU_LANE] = pred ? 1 : 0; emu_sync();
    unsigned long long m = 0; for (int i = 0; i < 64; i++) if (emu_xchg[i]) m |= 1ull << i;
    emu_sync(); return m;
}
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() do { __label__ emu_wb; emu_wb: emu_site[threadIdx.x] = &&emu_wb; emu_note(&&emu_wb); emu_sync(); } while (0)
static inline uint32_t __builtin_amdgcn_ubfe(uint32_t x, uint32_t off, uint32_t width) { off &= 31; width &= 31; return width ? (x >> off) & ((1u << width) - 1u) : 0u; }
/* v_perm_b32: result byte i = byte (selector byte i) of the 8 bytes {S0 (4..7), S1 (0..3)}; 0x0C -> 0x00 */
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    const uint64_t both = ((uint64_t)s0 << 32) | s1; uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t c = (sel >> (8 * i)) & 0xFF;
        uint32_t b;
        if (c <= 7) b = (uint32_t)(both >> (8 * c)) & 0xFF;
        else if (c <= 11) __builtin_trap();                                  /* sign replication: not used by the kernels */
        else if (c == 0x0C) b = 0; else b = 0xFF;                           /* 12: 0x00, >= 13: 0xFF */
        r |= b << (8 * i);
    }
    return r;
}
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31)); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   /* callers only pass wave-uniform values */
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return v ? __builtin_ctzll((unsigned long long)v) + 1 : 0; }
#ifndef __clang__
#define __builtin_nontemporal_load(p) (*(p))
#endif
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicExch(uint32_t* p, uint32_t v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline uint32_t atomicAnd(uint32_t* p, uint32_t v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
