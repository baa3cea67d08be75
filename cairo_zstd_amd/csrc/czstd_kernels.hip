/*
 * czstd_kernels.hip — CDNA4 (gfx950) kernels of the zstd frame/block decoder.
 *
 * cz_decode_frames_kernel: ONE 64-lane wavefront (= one workgroup) per frame, a
 * persistent grid that pulls frames from an atomic work counter.  Blocks of one frame are
 * chained (Treeless literals, Repeat FSE modes, offset history, window reach-back:
 * src/decoding/scratch.cairo:11-19), so the frame is the unit of parallelism and all carried
 * state lives in the workgroup's LDS: Huffman table (4 KiB), LL/OF/ML FSE tables (5 KiB),
 * offset history.  Per block:
 *     lane 0      parses headers / table descriptions from an LDS-staged copy of the bytes
 *     lanes 0..2  build the three FSE decoding tables side by side (only for blocks the pre-pass did not take)
 *     all lanes   fill the Huffman table; copy / fill Raw and RLE payloads (16 B per lane)
 *     all lanes   decode the huff0 streams: each stream cut into up to 16 bit ranges, one lane per range
 *                 (self-synchronising: counting pass, start fix-up, writing pass)
 *     lane 0      runs the interleaved LL/OF/ML FSE state machines, 64 sequences at a time — unless
 *                 cz_chain_kernel left per-sequence records (czstd_chain.hip) — then all 64 lanes
 *                 execute those sequences: a DPP scan resolves repeat offsets, wave prefix sums give
 *                 every sequence its literal and output offsets, short chunks are assembled in LDS,
 *                 matches are resolved in dependency rounds (ballot + first-undone watermark), long
 *                 copies are done cooperatively.
 * cz_dict_setup_kernel parses a dictionary with the same table builders.  Compiled a second time with CZ_EXEC_ONLY (in
 * namespace czx) this file is cz_execute_frames_kernel: the same block walk and record-driven execution without any decoder.
 *
 * Semantics follow the reference (NethermindEth/cairo_zstd) line by line where it matters;
 * each device function cites the reference file:line it restates.  Error codes mirror the
 * reference's enum leaves (cairo_zstd_amd_status.h) in the reference's order of detection.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "czstd_types.h"

#define LANE ((int)(threadIdx.x & 63u))   /* lane of the wavefront (cz_huf_kernel, which shares this file's helpers, has workgroups of several waves) */
#define CZ_NOINLINE __attribute__((noinline))
/* Pointers to global memory say so in their type.  A generic pointer that crosses a function that is
 * not inlined (or sits in a struct) makes the compiler emit flat_* instructions, which count against
 * BOTH vmcnt and lgkmcnt: every LDS wait then also waits for all global loads in flight. */
#if defined(__HIP_DEVICE_COMPILE__)
#define CZ_GLOBAL __attribute__((address_space(1)))
#else
#define CZ_GLOBAL            /* host pass of the same translation unit (and the CPU emulator): the device code is only parsed */
#endif
typedef CZ_GLOBAL const uint8_t* cz_gcptr;
typedef CZ_GLOBAL uint8_t* cz_gptr;
typedef CZ_GLOBAL uint4* cz_gptr4;
typedef CZ_GLOBAL const uint64_t* cz_gcptr64;
typedef CZ_GLOBAL uint16_t* cz_gptr16;
/* Values that are the same in every lane (read from the LDS broadcast slots, or produced by a
 * cross-lane broadcast) are pinned to scalar registers: keeps the VGPR budget for per-lane work
 * and lets the scalar unit do the uniform arithmetic. */
__device__ static inline uint32_t cz_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ static inline int32_t cz_unii(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ static inline uint64_t cz_uni64(uint64_t v) { return ((uint64_t)cz_uni((uint32_t)(v >> 32)) << 32) | cz_uni((uint32_t)v); }
/* Do cz_wexec_kernel and cz_execute_frames_kernel share this batch's frames?  Only when far offsets outweigh near ones (what
   cz_chain_kernel summed from the blocks' offset-code tables): a workgroup per frame pays when its waves seldom wait for each
   other's bytes; on near-offset data it is no faster than one wave and holds a whole CU. */
#ifndef CZ_EXEC_ONLY
__device__ static inline int cz_wx_side_by_side(const cz_batch_args& a) {
    return a.wx_list != nullptr && (a.wx_force ? a.wx_force == 1u : a.chain_top[6] > a.chain_top[5]);
}
/* Which build of cz_execute_frames_kernel runs this batch: 8 waves per SIMD (64 registers) when its offsets are near — then the
   kernel waits on its own recent stores rather than on random reads from HBM, and twice the waves hide twice as much of that —
   AND its code tables give next to no weight (under 1 sequence in 256) to literal runs above 8 or matches above 16 bytes, which
   the fast loop does not take (the general loop spills at 64 registers); else 4 waves per SIMD.  Both builds are launched; the
   other one returns at once. */
__device__ static inline uint32_t cz_exec_variant(const cz_batch_args& a) {
    if (a.exec_variant_force) return a.exec_variant_force;
    if (!a.chain_top) return 4u;
    const unsigned long long nearq = a.chain_top[5], farq = a.chain_top[6], longq = a.chain_top[7];
    return nearq > farq && longq * 256ull < nearq + farq ? 8u : 4u;
}
/* ... and on a near-offset batch that the 4-waves build runs: do the batch's LARGE frames (CZ_PRE_WXBIG, a few hundred at most) go to
   cz_wexec_kernel, all the others to cz_execute_frames_kernel?  Such a batch ends when its largest frames do, and a large frame
   shares its SIMD, the L2 and the memory system with 4 095 other waves there — 1.7 x its time alone; on cz_wexec_kernel it has a
   CU and its window to itself (corpus-like mix: execute stage 7.9 -> 5.6 ms, profiles/r4/NOTES.md).  Only where the chip IS that
   full and the large frames are few: 2 048 frames or more, at most one in sixteen of them large. */
__device__ static inline int cz_wx_big_only(const cz_batch_args& a) {
    const uint32_t nbig = a.scan_ctl[210];
    return a.wx_list != nullptr && a.wx_force != 2u && !cz_wx_side_by_side(a) && nbig != 0u && a.n >= CZ_WX_BIG_MIN_FRAMES && (uint64_t)nbig * CZ_WX_BIG_SHARE <= a.n && cz_exec_variant(a) == 4u;
}
/* Side by side (far-offset batches), cz_wexec_kernel takes half of the CUs and cz_execute_frames_kernel the others.  A workgroup of
   cz_wexec_kernel fills a CU (152 KB of LDS, 16 x 128 registers), so it can only ever start on a CU that holds no wave of the other
   kernel: round 4 got that by launching it first and the other kernel behind an event — which HIP does not promise (dispatch order
   is undefined); submitted second it found every CU held by persistent waves and did nothing.  Now cz_wexec_kernel's workgroups
   count themselves in (scan_ctl[213]; args.wx_cus of them stay, the launch has more), and a wave of cz_execute_frames_kernel that finds
   itself on an EVEN CU (s_getreg HW_REG_HW_ID, cu_id bit 0) waits up to ~30 us for them to be all in place and leaves if they are not —
   its launch has twice the waves the other half of the chip holds.  Dispatched first (the usual case), cz_wexec_kernel has its CUs before
   the first wave of the other kernel asks, nobody leaves, and the placement is the dispatcher's, which measured 0.1 ms faster on
   config 4a than any split by id (profiles/r5/NOTES.md); dispatched second, it finds the even CUs free.  On this part every XCD has
   4 shader engines x 8 active CUs with ids 0..8, 128 even and 128 odd (scripts/micro/census.hip).  1: even, 0: odd; the CPU emulator
   runs the kernels one after the other and has no CUs: 2. */
__device__ static inline uint32_t cz_cu_side() {
#if defined(CZ_EMU) || !defined(__HIP_DEVICE_COMPILE__)
    return 2u;
#else
#ifndef CZ_CU_SIDE_RULE
#define CZ_CU_SIDE_RULE 0
#endif
    const uint32_t hw = __builtin_amdgcn_s_getreg((31u << 11) | (0u << 6) | 4u);    /* HW_REG_HW_ID (4): cu_id [11:8], se_id [15:13] */
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3u << 11) | (0u << 6) | 20u);   /* HW_REG_XCC_ID (20): [3:0] */
    const uint32_t cu = (hw >> 8) & 15u, se = (hw >> 13) & 7u;
    (void)xcc; (void)se; (void)cu;
    return CZ_CU_SIDE_RULE == 0 ? (cu & 1u) ^ 1u : CZ_CU_SIDE_RULE == 1 ? (xcc & 1u) ^ 1u : CZ_CU_SIDE_RULE == 2 ? (xcc < 4u ? 1u : 0u)
         : CZ_CU_SIDE_RULE == 3 ? (se & 1u) ^ 1u : ((cu >> 1) & 1u) ^ 1u;
#endif
}
/* Agent-scope hand-off between kernels that run at the same time (the large blocks' chains -> cz_wexec_kernel's early launch).
   Producer: its stores, s_waitcnt vmcnt(0), CZ_RELEASE_AGENT (writes the XCD's L2 back; the explicit wait after it is not
   optional: the compiler drops its own when the scoreboard looks empty), then the flag with CZ_ST_AGENT.  Consumer: polls the flag
   with CZ_LD_AGENT (bypasses its CU's L1), then CZ_ACQUIRE_AGENT (invalidates that L1; waits for it), a workgroup barrier for the
   other waves, then plain loads. */
#ifdef CZ_EMU
#define CZ_RELEASE_AGENT() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define CZ_ACQUIRE_AGENT() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define CZ_ST_AGENT(p, v) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
#define CZ_LD_AGENT(p) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#else
#define CZ_RELEASE_AGENT() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define CZ_ACQUIRE_AGENT() do { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define CZ_ST_AGENT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CZ_LD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
/* Hands frame f to cz_decode_frames_kernel (fallback_list).  Several kernels may hand the same frame back — a huff0 kernel that
   met something irregular in one of its sections, the execute kernel that then finds the frame without literals, cz_wexec_kernel
   that still has it on its list — so the entry is made by whoever sets CZ_PRE_LISTED first: a frame is listed ONCE, the list never
   holds more than n entries, and no frame is decoded twice at the same time.  One lane calls it. */
__device__ static inline void cz_list_fallback(const cz_batch_args& a, uint32_t f) {
    if (!a.fallback_list) return;
    if (a.frame_pre && (atomicOr(&a.frame_pre[f], CZ_PRE_LISTED) & CZ_PRE_LISTED)) return;
    const uint32_t i = atomicAdd(a.fallback_count, 1u);
    if (i < a.n) a.fallback_list[i] = f;
}
#endif
/* Diagnostic build only (-DCZ_PROFILE, csrc/Makefile target `prof`): lane 0 accumulates
 * s_memtime deltas per phase into sh.prof[] and adds them to args.prof[] at the end of each
 * frame.  No stamp executes in the product build. */
#ifdef CZ_PROFILE
#define CZ_PROF_DECL unsigned long long cz_t_ = 0
#define CZ_PROF_T0() do { if (LANE == 0) cz_t_ = __builtin_amdgcn_s_memtime(); } while (0)
#define CZ_PROF_ACC(idx) do { if (LANE == 0) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); sh.prof[idx] += n_ - cz_t_; cz_t_ = n_; } } while (0)
#define CZ_PROF_CNT(idx) do { if (LANE == 0) sh.prof[idx] += 1; } while (0)
#else
#define CZ_PROF_CNT(idx) do { } while (0)
#define CZ_PROF_DECL
#define CZ_PROF_T0() do { } while (0)
#define CZ_PROF_ACC(idx) do { } while (0)
#endif
enum { CZ_P_HDR = 0, CZ_P_HUFBUILD, CZ_P_HUFDEC, CZ_P_SEQTAB, CZ_P_RING, CZ_P_CHAIN, CZ_P_EXTRACT, CZ_P_LITCOPY, CZ_P_MATCH, CZ_P_RAWRLE, CZ_P_OTHER,
       CZ_P_HUF_SPEC, CZ_P_HUF_SYNC, CZ_P_HUF_WRITE,                  /* sub-phases of CZ_P_HUFDEC (counted in both) */
       CZ_P_N_FAST, CZ_P_N_GENERAL, CZ_P_N_ROUNDS, CZ_P_N_BIG, CZ_P_COUNT };   /* counts: chunks on the LDS path / the general path, dependency rounds and wave-wide copies of the general path */
#define CZX_FALLBACK 0x7FF00001   /* internal: cz_execute_frames_kernel met a block it has no pre-pass results for; the frame goes to cz_decode_frames_kernel */
#define CZ_RING_BYTES 2048u
#define CZ_RING_BLOCK 1024u
#define CZ_RING_NEED 768u   /* >= 64 sequences x 89 bits */
/* The chunk buffer: a chunk's output is assembled in LDS when it is at most CZ_OBUF_BYTES long and no literal run or match of the
   chunk is longer than CZ_OBUF_MAXLEN.  cz_execute_frames_kernel at 4 waves per SIMD has the LDS for 3 KiB (+ as much again for the
   chunk's literals): chunks of real encoder output average ~900 bytes, and every chunk beyond the buffer pays a memory round
   trip per dependency round (profiles/r4/NOTES.md). */
#undef CZ_OBUF_BYTES
#undef CZ_OBUF_MAXLEN
#if defined(CZ_EXEC_ONLY) && CZ_EXEC_WAVES == 4 && !defined(CZ_EXP_SMALL_OBUF)
#define CZ_OBUF_BYTES 3072u
#ifdef CZ_EXP_MAXLEN
#define CZ_OBUF_MAXLEN CZ_EXP_MAXLEN
#else
#define CZ_OBUF_MAXLEN 128u
#endif
#else
#define CZ_OBUF_BYTES 1024u
#define CZ_OBUF_MAXLEN 64u
#endif


/* ------------------------------------------------------------------ LDS layout */
struct CzBroadcast {
    int32_t  err; uint32_t detail;
    /* frame header */
    uint32_t hdr_len, has_checksum; uint64_t window_size; uint64_t d0, d1;
    /* block header */
    uint32_t btype, bsize, blast;
    /* literals section */
    uint32_t lit_type, regen, nstreams, lit_total;      /* lit_total = header + body bytes */
    uint32_t stream_off[4], stream_len[4];              /* relative to the block start */
    uint32_t wt_nprobs, wt_log, wt_fse_bytes;   /* description of the Huffman-weight FSE table between the two phases of the tree parse */
    uint32_t huf_fill, huf_nsym, huf_last_w;   /* huf_last_w: the implied weight of the last symbol (huff0_decoder.cairo:359-372) */
    uint32_t st_count[4], st_flags[4];
    /* sequences section */
    int32_t  seq_hdr_err; uint32_t nseq, seq_modes, seq_body_off;
    uint32_t build_mask, nprobs[3], acc_log[3], bitstream_off;
    /* per chunk */
    uint32_t chunk_cnt; int32_t chunk_err;
};

/* LDS per workgroup (= per frame in flight).  The carried state of the reference's
 * DecoderScratch (scratch.cairo:11-19) that must survive from block to block stays resident:
 * three FSE tables, RLE symbols, offset history.  Everything else is phase-local and shares
 * region `a`:
 *   T1 parse literals section / Huffman weights   (stage, probs0, counters0, wtab)
 *   T2 Huffman table, live while the literal streams decode        (huf; the streams themselves are read
 *      straight from global memory into per-lane register windows)
 *   T3 parse + build the sequence tables                           (stage, probs, counters)
 *   T4 sequence decode                                             (bit ring, chain records)
 * The Huffman table is the only carried item that does not stay in LDS: when a block that
 * created one is not the last block it is spilled to a 4 KiB global slot and re-read by
 * Treeless blocks (literals_section_decoder.cairo:82-86). */
#ifdef CZ_EXEC_ONLY
/* cz_execute_frames_kernel (this file compiled a second time, in namespace czx, with CZ_EXEC_ONLY): frames whose FSE chains
 * and Huffman literals were done by the pre-pass need no decoding tables, no bit ring and no Huffman table: what is left is
 * the chunk buffer, the block-head stage of the header parser and the broadcast slots (1.9 KB + 1 280 B of state->code maps). */
struct CzShared {
    uint32_t hist[3]; int32_t fse_rle[3]; uint8_t fse_log[3]; uint8_t huf_max_bits;
    union {
        uint16_t huf[512];                                              /* cz_xxh64_frame stages 2 x 512 B here */
        struct { uint8_t stage[512]; } t1;
        struct { __attribute__((aligned(16))) uint8_t obuf[CZ_OBUF_BYTES + 16 + 64 + 16];
                 __attribute__((aligned(16))) uint8_t lstage[CZ_OBUF_BYTES + 32]; } t4;   /* lstage: the literals of the chunk in hand (cz_copy_long_runs) */
    } a;
    struct { struct { uint32_t llml[96]; } c; } b;
    CzBroadcast bc;
    uint32_t frame_idx;
    uint32_t rec_next, rec_misses;      /* the chunk cz_sequences_rec_fast stopped at; chunks of the frame in flight it has left to the general form */
    uint32_t dict_lag[2];
    uint32_t dict_ptr[2], dict_len[2];
#ifdef CZ_PROFILE
    unsigned long long prof[CZ_P_COUNT];
#endif
};
#else
struct CzShared {
    uint32_t hist[3]; int32_t fse_rle[3]; uint8_t fse_log[3]; uint8_t huf_max_bits;
    union {
        uint16_t huf[2048];
        struct { uint8_t stage[512]; int16_t probs0[256]; uint16_t counters0[256]; uint32_t wtab[512]; uint32_t rank_cnt[16], rank_idx[16]; } t1;
        struct { uint8_t stage[512]; int16_t probs[3][256]; uint16_t counters[3][256]; } t3;
        struct { __attribute__((aligned(16))) uint8_t mirror[16]; uint8_t ring[CZ_RING_BYTES]; int32_t rec_pos[64]; uint32_t rec_st[64];
                 __attribute__((aligned(16))) uint8_t obuf[CZ_OBUF_BYTES + 16 + 64 + 16];   /* + one dump byte per lane (+ 7: cz_fast_group) */   /* mirror[8..15] == ring[2040..2047] */
                 __attribute__((aligned(16))) uint8_t lstage[CZ_OBUF_BYTES + 32]; } t4;
    } a;
    struct {
        struct { __attribute__((aligned(4))) uint8_t hbits[264]; uint16_t sym_base[264]; uint32_t llml[96]; } c;   /* llml: [0..35] LL base | bits<<24, [40..92] ML */
    } b;
    CzBroadcast bc;
    uint32_t frame_idx;
    uint32_t rec_next, rec_misses;      /* the chunk cz_sequences_rec_fast stopped at; chunks of the frame in flight it has left to the general form */
    uint32_t dict_lag[2];               /* cz_device_frame_state.dict_lag of the frame in flight (lo, hi) */
    uint32_t dict_ptr[2], dict_len[2];  /* DecodeBuffer.dict_content of the frame in flight: kept here, not in registers — only the rare
                                           dictionary arm of the match copy reads them */
#ifdef CZ_PROFILE
    unsigned long long prof[CZ_P_COUNT];
#endif
};
/* The workgroup's LDS lives at module scope: every function, inlined or not, then addresses it in the
 * LDS address space (ds_* instructions, lgkmcnt only).  Passing it by reference through a function
 * that is not inlined turns the accesses into flat_* instructions, whose waits also cover every
 * outstanding global load. */
#endif
__shared__ CzShared sh;
CZ_DYNAMIC_LDS(cz_dyn_lds);                                             /* CZ_FSE_LDS_BYTES: the three FSE decoding tables */
#define CZ_FSE_LL (cz_dyn_lds)          /* 512 entries */
#define CZ_FSE_ML (cz_dyn_lds + 512)    /* 512 entries */
#define CZ_FSE_OF (cz_dyn_lds + 1024)   /* 256 entries */
__device__ static inline uint32_t* cz_fse_table(int t) { return t == 0 ? CZ_FSE_LL : (t == 1 ? CZ_FSE_OF : CZ_FSE_ML); }

/* sequence_section_decoder.cairo:299-345 / :347-395 */
__device__ static const uint32_t CZ_LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
__device__ static const uint8_t  CZ_LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__device__ static const uint32_t CZ_ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
__device__ static const uint8_t  CZ_ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
/* predefined distributions, sequence_section_decoder.cairo:418-455, :494-524, :562-616 */
__device__ static const int8_t CZ_LL_DEFAULT[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__device__ static const int8_t CZ_OF_DEFAULT[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
__device__ static const int8_t CZ_ML_DEFAULT[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};

/* math.cairo:266-271: 1-based index of the highest set bit */
__device__ static inline uint32_t cz_hbs(uint32_t v) { return v ? 32u - (uint32_t)__clz((int)v) : 0u; }

__device__ static inline void cz_init_llml();
/* ------------------------------------------------------------------ wave helpers */
/* Cross-lane moves on the VALU (DPP) instead of the LDS crossbar: row_shr:n within rows of 16 lanes,
 * then row_bcast:15 / row_bcast:31 carry the row totals upwards — the gfx9 wave64 scan idiom.  A lane
 * without a source (or outside the row mask) gets `old`. */
template <int CTRL, int ROWMASK> __device__ static inline uint32_t cz_dpp(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROWMASK, 0xF, false);
}
#define CZ_DPP_SHR1 0x111
#define CZ_DPP_SHR2 0x112
#define CZ_DPP_SHR4 0x114
#define CZ_DPP_SHR8 0x118
#define CZ_DPP_WAVE_SHR1 0x138
#define CZ_DPP_BCAST15 0x142
#define CZ_DPP_BCAST31 0x143
__device__ static inline uint32_t cz_wave_incl_scan(uint32_t v) {
    v += cz_dpp<CZ_DPP_SHR1, 0xF>(0, v); v += cz_dpp<CZ_DPP_SHR2, 0xF>(0, v);
    v += cz_dpp<CZ_DPP_SHR4, 0xF>(0, v); v += cz_dpp<CZ_DPP_SHR8, 0xF>(0, v);
    v += cz_dpp<CZ_DPP_BCAST15, 0xA>(0, v); v += cz_dpp<CZ_DPP_BCAST31, 0xC>(0, v);
    return v;
}
__device__ static inline uint32_t cz_readlane(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
/* The lanes of a wave execute an instruction together, so the LDS writes of one instruction are all done before those of the next
 * begin; the CPU emulator of tests/emu runs lanes as threads and needs a barrier where the kernels rely on that. */
#ifdef CZ_EMU
#define CZ_LOCKSTEP() cz_wave_sync()
#else
#define CZ_LOCKSTEP() asm volatile("" ::: "memory")   /* no instruction; the compiler must keep the program order of the LDS writes around it (it sees one lane, for which they never alias) */
#endif
/* A workgroup is ONE wave (CZ_WG_THREADS == 64): its LDS and vector-memory instructions execute in
 * program order, so making one lane's write visible to another lane's later read needs no wait and
 * no s_barrier — only that the compiler keeps the order. */
__device__ static inline void cz_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

/* all-lane copy, 16 B per lane per step once dst is 16-byte aligned.  Both pointers are
 * global memory: say so, so that the loop uses global_load / global_store (not flat_*). */
__device__ static void cz_coop_copy(uint8_t* dst_, const uint8_t* src_, uint64_t n) {
    cz_gptr dst = (cz_gptr)dst_; cz_gcptr src = (cz_gcptr)src_;
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst_ & 15u)) & 15u);
    if (head > n) head = (uint32_t)n;
    if ((uint32_t)LANE < head) dst[LANE] = src[LANE];
    dst += head; src += head; n -= head;
    const uint64_t nvec = n >> 4;
    uint64_t i = (uint64_t)LANE;
    /* 8 KiB per wave in flight: eight independent 16-byte loads per lane before the first store,
       so the loop is bound by bandwidth rather than by one HBM round trip per KiB */
    for (; i + 7 * 64 < nvec; i += 8 * 64) {
        uint4 v0, v1, v2, v3, v4, v5, v6, v7;                          /* source may be unaligned */
        __builtin_memcpy(&v0, src + 16 * i, 16); __builtin_memcpy(&v1, src + 16 * (i + 64), 16);
        __builtin_memcpy(&v2, src + 16 * (i + 128), 16); __builtin_memcpy(&v3, src + 16 * (i + 192), 16);
        __builtin_memcpy(&v4, src + 16 * (i + 256), 16); __builtin_memcpy(&v5, src + 16 * (i + 320), 16);
        __builtin_memcpy(&v6, src + 16 * (i + 384), 16); __builtin_memcpy(&v7, src + 16 * (i + 448), 16);
        *(cz_gptr4)(dst + 16 * i) = v0; *(cz_gptr4)(dst + 16 * (i + 64)) = v1; *(cz_gptr4)(dst + 16 * (i + 128)) = v2; *(cz_gptr4)(dst + 16 * (i + 192)) = v3;
        *(cz_gptr4)(dst + 16 * (i + 256)) = v4; *(cz_gptr4)(dst + 16 * (i + 320)) = v5; *(cz_gptr4)(dst + 16 * (i + 384)) = v6; *(cz_gptr4)(dst + 16 * (i + 448)) = v7;
    }
    for (; i + 3 * 64 < nvec; i += 4 * 64) {                            /* medium copies: four in flight */
        uint4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, src + 16 * i, 16); __builtin_memcpy(&v1, src + 16 * (i + 64), 16);
        __builtin_memcpy(&v2, src + 16 * (i + 128), 16); __builtin_memcpy(&v3, src + 16 * (i + 192), 16);
        *(cz_gptr4)(dst + 16 * i) = v0; *(cz_gptr4)(dst + 16 * (i + 64)) = v1; *(cz_gptr4)(dst + 16 * (i + 128)) = v2; *(cz_gptr4)(dst + 16 * (i + 192)) = v3;
    }
    for (; i < nvec; i += 64) {
        uint4 v; __builtin_memcpy(&v, src + 16 * i, 16);
        *(cz_gptr4)(dst + 16 * i) = v;
    }
    for (uint64_t t = (nvec << 4) + (uint64_t)LANE; t < n; t += 64) dst[t] = src[t];
}
__device__ static void cz_coop_fill(uint8_t* dst_, uint8_t byte, uint64_t n) {
    cz_gptr dst = (cz_gptr)dst_;
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst_ & 15u)) & 15u);
    if (head > n) head = (uint32_t)n;
    if ((uint32_t)LANE < head) dst[LANE] = byte;
    dst += head; n -= head;
    const uint32_t w = 0x01010101u * byte;
    uint4 v; v.x = w; v.y = w; v.z = w; v.w = w;
    const uint64_t nvec = n >> 4;
    for (uint64_t i = (uint64_t)LANE; i < nvec; i += 64) *(cz_gptr4)(dst + 16 * i) = v;
    for (uint64_t i = (nvec << 4) + (uint64_t)LANE; i < n; i += 64) dst[i] = byte;
}

/* ------------------------------------------------------------------ bit readers */
/* Reversed reader (bit_reader_reverse.cairo:44-275), one per lane, over global memory.
 * `remaining` is the reference's bits_remaining and may go negative; reads past the start
 * return zero bits (:147-159).  buf holds unread bits MSB-aligned. */
struct CzRBits { const uint8_t* base; int32_t bytes_left; uint64_t buf; int32_t avail; int32_t remaining; };

__device__ static inline void cz_rb_init(CzRBits& r, const uint8_t* base, uint32_t len) {
    r.base = base; r.bytes_left = (int32_t)len; r.buf = 0; r.avail = 0; r.remaining = (int32_t)len * 8;
}
__device__ static inline void cz_rb_refill(CzRBits& r) {
    if (r.avail <= 32 && r.bytes_left > 0) {
        if (r.bytes_left >= 4) {
            const uint8_t* p = r.base + r.bytes_left - 4;
            uint32_t w = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
            r.bytes_left -= 4;
            r.buf |= (uint64_t)w << (32 - r.avail); r.avail += 32;
        } else {
            while (r.bytes_left > 0) { uint8_t b = r.base[--r.bytes_left]; r.buf |= (uint64_t)b << (56 - r.avail); r.avail += 8; }
        }
    }
}
/* get_bits(n), n <= 32 (bit_reader_reverse.cairo:129-172) */
__device__ static inline uint32_t cz_rb_get(CzRBits& r, uint32_t n) {
    if (n == 0) return 0;
    cz_rb_refill(r);
    uint32_t v = (uint32_t)(r.buf >> (64 - n));
    r.buf <<= n; r.avail -= (int32_t)n; r.remaining -= (int32_t)n;
    return v;
}
/* zero padding up to and including the first 1 bit; > 8 reads = ExtraPadding
 * (literals_section_decoder.cairo:190-207, sequence_section_decoder.cairo:46-64,
 *  huff0_decoder.cairo:206-225).  Returns 1 on ExtraPadding. */
__device__ static inline int cz_rb_skip_padding(CzRBits& r) {
    int skipped = 0;
    for (;;) { uint32_t v = cz_rb_get(r, 1); skipped++; if (v == 1 || skipped > 8) break; }
    return skipped > 8;
}

/* Forward LSB-first reader (bit_reader.cairo:18-110) used by ONE lane for table
 * descriptions.  Bytes come from the LDS stage when inside it, else from global memory. */
struct CzFBits { cz_gcptr g; uint32_t len; const uint8_t* stage; uint32_t stage_lo, stage_hi; uint32_t idx; };
__device__ static inline uint32_t cz_fb_byte(const CzFBits& f, uint32_t i) {
    return (i >= f.stage_lo && i < f.stage_hi) ? f.stage[i - f.stage_lo] : f.g[i];
}
/* returns 0 ok / 1 not enough bits (bit_reader.cairo:42-44).  n <= 24 */
__device__ static inline int cz_fb_get(CzFBits& f, uint32_t n, uint32_t* out) {
    if (f.len * 8u - f.idx < n) return 1;
    uint32_t b = f.idx >> 3, sh = f.idx & 7, v = 0;
    for (uint32_t k = 0; k * 8 < sh + n; k++) v |= cz_fb_byte(f, b + k) << (8 * k);
    *out = (v >> sh) & ((1u << n) - 1u);
    f.idx += n; return 0;
}

/* ------------------------------------------------------------------ FSE tables */
/* packed entry (fse_decoder.cairo:49-53 plus what the sequence chain needs in ONE lookup):
 *   [6:0]  extra bits of the symbol's LL/ML/OF code (<= 31; 2 bits of headroom)
 *   [12:7] num_bits (<= 9; 2 bits of headroom)      [13] code out of range (LL>=36, ML>=53, OF>=32)
 *   [22:14] base_line                                [31:24] symbol
 * The headroom lets the chain add the three entries of a sequence and read the summed extra
 * bits and summed state bits straight out of the sum. */
#define CZ_FSE_SYM(e) ((e) >> 24)
#define CZ_FSE_NB(e) (((e) >> 7) & 0x3Fu)
#define CZ_FSE_XB(e) ((e) & 0x7Fu)
#define CZ_FSE_INV(e) (((e) >> 13) & 1u)
#define CZ_FSE_BASE(e) (((e) >> 14) & 0x1FFu)
#define CZ_FSE_PACK(sym, nb, base) (((uint32_t)(sym) << 24) | ((uint32_t)(nb) << 7) | ((uint32_t)(base) << 14))
/* extra-bits / validity part of an entry for symbol s of table kind (0 LL, 1 OF, 2 ML, 3 none) */
__device__ static inline uint32_t cz_fse_code_bits(const uint32_t* llml, uint32_t kind, uint32_t s) {
    if (kind == 0) return s < 36 ? (llml[s] >> 24) : (1u << 13);
    if (kind == 1) return s < 32 ? s : (1u << 13);
    if (kind == 2) return s < 53 ? (llml[40 + s] >> 24) : (1u << 13);
    return 0;
}

#ifndef CZ_EXEC_ONLY
/* read_probabilities (fse_decoder.cairo:258-368); probs -> LDS.  One lane. */
/* (the body is force-inlined where the callers' pointers are known to be LDS: a generic pointer that crosses a function call makes
   every access a flat_* instruction, which waits for global memory too — cz_chain_kernel's table set-up calls the _inl form) */
__device__ static inline __attribute__((always_inline)) int cz_fse_read_probs_inl(CzFBits& br, uint32_t max_log, int16_t* probs, uint32_t* nprobs,
                                        uint32_t* acc_log, uint32_t* bytes_read, int unsupported_above, uint32_t cap = 256) {
    uint32_t v;
    if (cz_fb_get(br, 4, &v)) return CZ_E_FSE_GETBITS;                  /* :265-268 */
    uint32_t log = 5 + v;                                               /* :270 */
    if (log > max_log) return CZ_E_FSE_ACC_LOG_TOO_BIG;                 /* :271 */
    if ((int)log > unsupported_above) return CZ_E_UNSUPPORTED;          /* DESIGN.md divergence D2 */
    uint32_t sum = 1u << log, counter = 0, n = 0;
    while (counter < sum) {                                             /* :281-346 */
        uint32_t max_rem = sum - counter + 1, bits = cz_hbs(max_rem);
        if (cz_fb_get(br, bits, &v)) return CZ_E_FSE_GETBITS;
        uint32_t low = ((1u << bits) - 1u) - max_rem, mask = (1u << (bits - 1)) - 1u, small = v & mask, value;
        if (small < low) { br.idx -= 1; value = small; }                /* return_bits(1) :303 */
        else if (v > mask) value = v - low;
        else value = v;
        int32_t prob = (int32_t)value - 1;
        if (n < cap) probs[n] = (int16_t)prob;
        n++;
        if (prob != 0) counter += prob > 0 ? (uint32_t)prob : 1u;
        else for (;;) {                                                 /* :322-340 */
            if (cz_fb_get(br, 2, &v)) return CZ_E_FSE_GETBITS;
            for (uint32_t k = 0; k < v; k++) { if (n < cap) probs[n] = 0; n++; }
            if (v != 3) break;
        }
    }
    if (counter != sum) return CZ_E_FSE_PROB_MISMATCH;                  /* :352 */
    if (n > 256) return CZ_E_FSE_TOO_MANY_SYMBOLS;                      /* :357 */
    *nprobs = n; *acc_log = log; *bytes_read = (br.idx + 7) >> 3;       /* :361-365 */
    return 0;
}
__device__ static __attribute__((noinline)) int cz_fse_read_probs(CzFBits& br, uint32_t max_log, int16_t* probs, uint32_t* nprobs,
                                        uint32_t* acc_log, uint32_t* bytes_read, int unsupported_above, uint32_t cap = 256) {
    return cz_fse_read_probs_inl(br, max_log, probs, nprobs, acc_log, bytes_read, unsupported_above, cap);
}
/* build_decoding_table (fse_decoder.cairo:156-256).  One lane per table; lanes 0..2 run it
 * side by side on different tables. */
__device__ static __attribute__((noinline)) void cz_fse_build(uint32_t* table, const int16_t* probs, uint32_t nprobs, uint32_t log, uint16_t* counters,
                                    const uint32_t* llml, uint32_t kind) {
    const uint32_t size = 1u << log;
    uint32_t neg = size;
    for (uint32_t s = 0; s < nprobs; s++) {                             /* :169-188 */
        counters[s] = 0;
        if (probs[s] == -1) { neg--; table[neg] = CZ_FSE_PACK(s, log, 0) | cz_fse_code_bits(llml, kind, s); }
    }
    uint32_t pos = 0; const uint32_t step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    for (uint32_t s = 0; s < nprobs; s++) {                             /* :190-226 */
        int32_t p = probs[s];
        for (int32_t j = 0; j < p; j++) {
            table[pos] = s << 24;
            do { pos = (pos + step) & mask; } while (pos >= neg);
        }
    }
    for (uint32_t i = 0; i < neg; i++) {                                /* :231-255, :377-400 */
        uint32_t s = table[i] >> 24, n = (uint32_t)probs[s], k = counters[s];
        counters[s] = (uint16_t)(k + 1);
        uint32_t m = 1u << (cz_hbs(n) - 1), slices = (m == n) ? n : m * 2;
        uint32_t dbl = slices - n, single = n - dbl, width = size >> (cz_hbs(slices) - 1), nb = cz_hbs(width) - 1, bl;   /* slices is a power of two */
        if (k < dbl) { bl = single * width + k * width * 2; nb += 1; }
        else bl = (k - dbl) * width;
        table[i] = CZ_FSE_PACK(s, nb, bl) | cz_fse_code_bits(llml, kind, s);
    }
}

/* ------------------------------------------------------------------ Huffman table */
/* read_weights + the serial half of build_table_from_weights
 * (huff0_decoder.cairo:159-319, :321-431).  Lane 0.  Leaves per-symbol code lengths in
 * sh.b.c.hbits[0..nsym) and first-cell indices in sh.b.c.sym_base[]; the table itself is filled by
 * all lanes afterwards (cz_huf_fill).  *bytes_used per :313-318. */
/* cz_fse_build by the whole wave (all 64 lanes; at most 64 symbols, table of at most 256 cells): the same three steps as
 * czc_fse_build_wave in czstd_chain.hip — lane = symbol for the "less than one" cells and the first rank of every symbol,
 * lane = step of the spreading walk, lane = cell for the entries, equal symbols of a chunk matched with six ballots —
 * with this kernel's 32-bit entries.  symof: `size` bytes of scratch, counters: 64 halfwords. */
#define CZ_PARSE_NEED_WTAB (-2)   /* cz_parse_sections / cz_huf_read_and_rank, phase 0: the weights' FSE table is described in bc.wt_*; build it and call phase 1 */
__device__ static inline void cz_fse_build_wave(uint32_t* table, const int16_t* probs, uint32_t nprobs, uint32_t log, uint8_t* symof, uint16_t* counters,
                                                 const uint32_t* llml, uint32_t kind) {
    const uint32_t size = 1u << log, mask = size - 1, lane = (uint32_t)LANE;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int32_t p = lane < nprobs ? (int32_t)probs[lane] : 0;
    const unsigned long long lowm = __ballot(p == -1);
    const uint32_t neg = size - (uint32_t)__popcll(lowm);
    if (p == -1) table[size - 1u - (uint32_t)__popcll(lowm & lt)] = CZ_FSE_PACK(lane, log, 0) | cz_fse_code_bits(llml, kind, lane);   /* :169-188 */
    counters[lane] = 0;
    const uint32_t cnt = p > 0 ? (uint32_t)p : 0u;
    const uint32_t cum = cz_wave_incl_scan(cnt) - cnt;
    for (uint32_t i = 4u * lane; i < size; i += 256u) *(uint32_t*)(symof + i) = 0u;
    cz_wave_sync();
    if (cnt && cum < size) symof[cum] = (uint8_t)lane;
    cz_wave_sync();
    {
        const uint32_t per = size >= 64u ? size >> 6 : 1u, b0 = lane * per;
        const int act = b0 < size;
        uint32_t run = 0, vals[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) { if (act && j < per) { const uint32_t v = symof[b0 + j]; run = v > run ? v : run; } vals[j] = run; }
        uint32_t inc = act ? run : 0u;
#define CZ_MAX_STEP(CTRL, RM) do { const uint32_t o_ = cz_dpp<CTRL, RM>(0u, inc); inc = o_ > inc ? o_ : inc; } while (0)
        CZ_MAX_STEP(CZ_DPP_SHR1, 0xF); CZ_MAX_STEP(CZ_DPP_SHR2, 0xF); CZ_MAX_STEP(CZ_DPP_SHR4, 0xF); CZ_MAX_STEP(CZ_DPP_SHR8, 0xF);
        CZ_MAX_STEP(CZ_DPP_BCAST15, 0xA); CZ_MAX_STEP(CZ_DPP_BCAST31, 0xC);
#undef CZ_MAX_STEP
        const uint32_t exc = cz_dpp<CZ_DPP_WAVE_SHR1, 0xF>(0u, inc);
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) if (act && j < per) symof[b0 + j] = (uint8_t)(vals[j] > exc ? vals[j] : exc);
    }
    cz_wave_sync();
    {
        const uint32_t step = (size >> 1) + (size >> 3) + 3u;          /* :190-226 */
        uint32_t running = 0;
        for (uint32_t j0 = 0; j0 < size; j0 += 64u) {
            const uint32_t j = j0 + lane, pos = (j * step) & mask;
            const int valid = j < size && pos < neg;
            const unsigned long long vm = __ballot(valid);
            if (valid) table[pos] = symof[running + (uint32_t)__popcll(vm & lt)];
            running += (uint32_t)__popcll(vm);
        }
    }
    cz_wave_sync();
    for (uint32_t i0 = 0; i0 < neg; i0 += 64u) {                        /* :231-255, :377-400 */
        const uint32_t i = i0 + lane;
        const int cell = i < neg;
        const uint32_t s = cell ? table[i] : 0u;
        unsigned long long same = __ballot(cell);
#pragma unroll
        for (uint32_t b = 0; b < 6; b++) { const int bit = (int)((s >> b) & 1u); const unsigned long long m = __ballot(bit); same &= bit ? m : ~m; }
        const uint32_t n = cell ? (uint32_t)probs[s] : 1u;
        const uint32_t k = (cell ? (uint32_t)counters[s] : 0u) + (uint32_t)__popcll(same & lt);
        cz_wave_sync();                                                 /* every lane has read counters[] */
        if (cell && (same >> lane) == 1ull) counters[s] = (uint16_t)(k + 1u);
        if (cell) {
            const uint32_t m = 1u << (cz_hbs(n) - 1), slices = (m == n) ? n : m * 2;
            const uint32_t dbl = slices - n, single = n - dbl, width = size >> (cz_hbs(slices) - 1);
            uint32_t nb = cz_hbs(width) - 1, bl;
            if (k < dbl) { bl = single * width + k * width * 2; nb += 1; }
            else bl = (k - dbl) * width;
            table[i] = CZ_FSE_PACK(s, nb, bl) | cz_fse_code_bits(llml, kind, s);
        }
        cz_wave_sync();
    }
}
/* the weights' FSE table between the phases of the tree parse: by the wave when it fits that builder, else by lane 0 as before */
__device__ static inline void cz_huf_weight_table() {
    const uint32_t np = cz_uni(sh.bc.wt_nprobs), lg = cz_uni(sh.bc.wt_log);
    if (np <= 64u && lg <= 8u) cz_fse_build_wave(sh.a.t1.wtab, sh.a.t1.probs0, np, lg, (uint8_t*)(sh.a.t1.wtab + 256), sh.a.t1.counters0, sh.b.c.llml, 3);
    else if (LANE == 0) cz_fse_build(sh.a.t1.wtab, sh.a.t1.probs0, np, lg, sh.a.t1.counters0, sh.b.c.llml, 3);
}
/* phase 0 stops (CZ_PARSE_NEED_WTAB) once an FSE-compressed description's probabilities are read; the caller has the table
   built (cz_huf_weight_table, all lanes) and calls phase 1, which goes on from there */
__device__ static __attribute__((noinline)) int cz_huf_read_and_rank(cz_gcptr g, uint32_t len, uint32_t stage_lo, uint32_t stage_hi,
                                           uint32_t goff, uint32_t* bytes_used, uint32_t* nsym_out, int phase) {
    /* g = block start, the tree description begins at block offset goff, len bytes available */
    if (len == 0) return CZ_E_HUF_SOURCE_EMPTY;                         /* :162 */
    CzFBits fb; fb.g = g; fb.stage = sh.a.t1.stage; fb.stage_lo = stage_lo; fb.stage_hi = stage_hi; fb.idx = 0; fb.len = 0;
    const uint32_t header = cz_fb_byte(fb, goff);
    uint8_t* w = sh.b.c.hbits; uint32_t nw = 0;
    if (header < 128) {                                                 /* :168-277 */
        const uint32_t fl = len - 1;
        if (header > fl) return CZ_E_HUF_NOT_ENOUGH_BYTES_FOR_WEIGHTS;  /* :171 */
        /* FSE description: reader positioned at block offset goff+1 */
        CzFBits br; br.g = g + goff + 1; br.len = fl; br.stage = sh.a.t1.stage; br.idx = 0;
        br.stage_lo = 0; br.stage_hi = 0;
        if (goff + 1 >= stage_lo && goff + 1 < stage_hi) { br.stage = sh.a.t1.stage + (goff + 1 - stage_lo); br.stage_lo = 0; br.stage_hi = stage_hi - (goff + 1); }
        uint32_t nprobs, log, fse_bytes;
        if (phase == 0) {
            int e = cz_fse_read_probs(br, 100, sh.a.t1.probs0, &nprobs, &log, &fse_bytes, 9);   /* :176 max_log 100; device cap 9 (D2) */
            if (e) return e;
            if (fse_bytes > header) return CZ_E_HUF_FSE_USED_TOO_MANY_BYTES; /* :181 */
            sh.bc.wt_nprobs = nprobs; sh.bc.wt_log = log; sh.bc.wt_fse_bytes = fse_bytes;
            return CZ_PARSE_NEED_WTAB;
        }
        nprobs = sh.bc.wt_nprobs; log = sh.bc.wt_log; fse_bytes = sh.bc.wt_fse_bytes; (void)nprobs;
        /* the weight bitstream (<= 127 bytes) is read from the LDS stage when it lies inside it: one dependent global
           load per refill otherwise */
        const uint32_t wlo = goff + 1 + fse_bytes, whi = goff + 1 + header;
        const uint8_t* wbase = (wlo >= stage_lo && whi <= stage_hi) ? (const uint8_t*)sh.a.t1.stage + (wlo - stage_lo) : (const uint8_t*)(g + wlo);
        CzRBits rb; cz_rb_init(rb, wbase, header - fse_bytes);          /* :190-202 */
        if (cz_rb_skip_padding(rb)) return CZ_E_HUF_EXTRA_PADDING;      /* :206-225 */
        uint32_t d1 = sh.a.t1.wtab[cz_rb_get(rb, log)];                    /* :227 */
        uint32_t d2 = sh.a.t1.wtab[cz_rb_get(rb, log)];                    /* :233 */
        for (;;) {                                                      /* :242-274 */
            if (nw < 260) w[nw] = (uint8_t)CZ_FSE_SYM(d1); nw++;
            d1 = sh.a.t1.wtab[CZ_FSE_BASE(d1) + cz_rb_get(rb, CZ_FSE_NB(d1))];
            if (rb.remaining <= -1) { if (nw < 260) w[nw] = (uint8_t)CZ_FSE_SYM(d2); nw++; break; }
            if (nw < 260) w[nw] = (uint8_t)CZ_FSE_SYM(d2); nw++;
            d2 = sh.a.t1.wtab[CZ_FSE_BASE(d2) + cz_rb_get(rb, CZ_FSE_NB(d2))];
            if (rb.remaining <= -1) { if (nw < 260) w[nw] = (uint8_t)CZ_FSE_SYM(d1); nw++; break; }
            if (nw > 255) return CZ_E_HUF_TOO_MANY_WEIGHTS;             /* :271 */
        }
        if (nw > 255) return CZ_E_HUF_TOO_MANY_WEIGHTS;                 /* u8 overflow panic at :458 */
        *bytes_used = 1 + header;                                       /* :204, :313-318 */
    } else {                                                            /* :278-311 direct weights (zstd nibble order, D1) */
        nw = header - 127;
        const uint32_t need = (nw + 1) >> 1;
        if (len - 1 < need) return CZ_E_HUF_NOT_ENOUGH_BYTES_IN_SOURCE; /* :289 */
        for (uint32_t i = 0; i < nw; i++) {
            uint32_t b = cz_fb_byte(fb, goff + 1 + (i >> 1));
            w[i] = (uint8_t)((i & 1) ? (b & 0xF) : (b >> 4));
        }
        *bytes_used = 1 + need;
    }
    /* build_table_from_weights :321-431 */
    uint32_t sum = 0, too_big = 0;
    {
        /* four weights per LDS read, no exit inside the loop (one dependent LDS round trip per weight made this loop as
           long as the weight decode itself); bytes beyond nw are masked */
        const uint32_t* w4 = (const uint32_t*)w;                        /* hbits is 4-byte aligned, 264 bytes long */
        for (uint32_t i = 0; i < nw; i += 4) {
            uint32_t v = w4[i >> 2];
            if (nw - i < 4) v &= 0xFFFFFFFFu >> (8 * (4 - (nw - i)));
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) { const uint32_t x = (v >> (8 * j)) & 0xFFu; too_big |= x > 11u; sum += x ? (1u << ((x - 1) & 31u)) : 0u; }
        }
    }
    if (too_big) return CZ_E_HUF_WEIGHT_TOO_BIG;                        /* :335 (the first failing check of the reference; nothing else is tested before it) */
    if (sum == 0) return CZ_E_HUF_MISSING_WEIGHTS;                      /* :351 */
    const uint32_t max_bits = cz_hbs(sum), left = (1u << max_bits) - sum;
    if (left == 0 || (left & (left - 1))) return CZ_E_HUF_LEFTOVER_NOT_POW2;            /* :359 */
    const uint32_t last_w = cz_hbs(left);
    sh.huf_max_bits = (uint8_t)max_bits;                             /* :383 */
    if (max_bits > 11) { sh.huf_max_bits = 0; return CZ_E_HUF_MAX_BITS_TOO_HIGH; }   /* :385; the reference leaves the too-large value in its (now unusable)
                                                                           table; here a resumed decoder must not index a 2^11-entry table with it */
    /* bit lengths, ranks and the first table index of every symbol: cz_huf_rank_wave, by all lanes, before cz_huf_fill */
    sh.bc.huf_last_w = last_w;
    *nsym_out = nw + 1;
    return 0;
}
/* all lanes: the cell-filling half (huff0_decoder.cairo:451-463).  entry = symbol | bits<<8 */
/* The rest of build_table_from_weights (huff0_decoder.cairo:374-450) by the whole wave — weights -> bit lengths, symbols per
 * length, and for every symbol the first index of its run in the decoding table (runs ordered by length, longest first,
 * symbols of one length in symbol order).  One lane did this in three dependent LDS loops over up to 256 symbols; here
 * lane = symbol, 64 at a time: lanes of equal length are matched with four ballots (the rank of a symbol among its length
 * within the chunk), earlier chunks are in cnt[].  hbits[] holds the weights on entry and the bit lengths on exit.
 * Scratch: sh.a.t1.rank_cnt / rank_idx (region `a` is the parse scratch until cz_huf_fill turns it into the table). */
__device__ static inline void cz_huf_rank_wave(uint32_t nsym) {
    const uint32_t max_bits = cz_uni(sh.huf_max_bits), last_w = cz_uni(sh.bc.huf_last_w), lane = (uint32_t)LANE;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* cnt = sh.a.t1.rank_cnt; uint32_t* idx = sh.a.t1.rank_idx;
    if (lane < 16) { cnt[lane] = 0; idx[lane] = 0; }
    cz_wave_sync();
    uint32_t bits[4], k[4];
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint32_t s = 64u * c + lane;
        const int in = s < nsym;
        const uint32_t wt = !in ? 0u : (s + 1 < nsym ? sh.b.c.hbits[s] : last_w);
        const uint32_t b = wt ? max_bits + 1 - wt : 0u;
        unsigned long long same = __ballot(b != 0);
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) { const int bit = (int)((b >> t) & 1u); const unsigned long long m = __ballot(bit); same &= bit ? m : ~m; }
        const uint32_t before = b ? cnt[b] : 0u;
        cz_wave_sync();                                                 /* every lane has read cnt[] */
        if (b && (same >> lane) == 1ull) cnt[b] = before + (uint32_t)__popcll(same);   /* the highest lane of the group */
        cz_wave_sync();
        bits[c] = b; k[c] = before + (uint32_t)__popcll(same & lt);
        if (in) sh.b.c.hbits[s] = (uint8_t)b;
    }
    if (lane == 0) for (uint32_t b = max_bits; b > 0; b--) idx[b - 1] = idx[b] + cnt[b] * (1u << (max_bits - b));   /* :414-429 */
    cz_wave_sync();
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {                                  /* :433-450 */
        const uint32_t s = 64u * c + lane;
        if (s < nsym && bits[c]) sh.b.c.sym_base[s] = (uint16_t)(idx[bits[c]] + (k[c] << (max_bits - bits[c])));
    }
    cz_wave_sync();
}
__device__ static __attribute__((noinline)) void cz_huf_fill(uint32_t nsym) {
    const uint32_t max_bits = cz_uni(sh.huf_max_bits);
    /* 64 symbols at a time: every lane fetches the length and the first index of one symbol (one LDS round trip for the
       chunk instead of one per symbol), then the coded symbols of the chunk are taken in turn from registers and the lanes
       write the symbol's run together */
    for (uint32_t s0 = 0; s0 < nsym; s0 += 64) {
        const uint32_t s = s0 + (uint32_t)LANE;
        const uint32_t bl = s < nsym ? sh.b.c.hbits[s] : 0u, bs = s < nsym ? sh.b.c.sym_base[s] : 0u;
        for (unsigned long long m = __ballot(bl != 0); m; m &= m - 1) {
            const int j = cz_unii(__ffsll((long long)m) - 1);
            const uint32_t b = cz_readlane(bl, j), base = cz_readlane(bs, j), len = 1u << (max_bits - b);
            const uint16_t e = (uint16_t)((s0 + (uint32_t)j) | (b << 8));
            for (uint32_t k = (uint32_t)LANE; k < len; k += 64) sh.a.huf[base + k] = e;
        }
    }
}
/* Entries of huf[] are symbol | length << 8; this adds, in bits 12..15, the length of the NEXT symbol when its whole
 * code lies inside the index bits too (a prefix code that matches the known bits is determined by them; 0 otherwise), so
 * that a pass that only counts can step two symbols at a time.  All lanes; huf[] complete. */
__device__ static inline void cz_huf_fill_multi() {
    const uint32_t mb = cz_uni(sh.huf_max_bits), size = 1u << mb, mask = size - 1;
    /* only the length bits 8..11 of other entries are read, and those never change: no ordering between the lanes is needed */
#pragma unroll 4
    for (uint32_t idx = (uint32_t)LANE; idx < size; idx += 64) {
        const uint32_t e = sh.a.huf[idx], l1 = (e >> 8) & 15u, l = (sh.a.huf[(idx << l1) & mask] >> 8) & 15u;
        sh.a.huf[idx] = (uint16_t)((e & 0x0FFFu) | ((l1 + l <= mb ? l : 0u) << 12));
    }
}
/* One huff0 stream, one lane (literals_section_decoder.cairo:183-243).  Writes at most `cap`
 * bytes to out but keeps counting.  flags: 1 ExtraPadding, 2 stream did not end exactly. */
__device__ static __attribute__((noinline)) void cz_huf_stream(cz_gcptr src, uint32_t len, cz_gptr out, uint32_t cap,
                                     uint32_t* count, uint32_t* flags) {
    CzRBits rb; cz_rb_init(rb, src, len);
    if (cz_rb_skip_padding(rb)) { *count = 0; *flags = 1; return; }    /* :190-207 */
    const uint32_t mb = sh.huf_max_bits;
    uint32_t n = 0;
    /* peek form of init_state/next_state (huff0_decoder.cairo:81-106): state = next mb bits.
       bits_remaining(ref) = rb.remaining - mb; loop while it is > -mb (:216-228). */
    while (rb.remaining > 0) {
        cz_rb_refill(rb);
        const uint32_t e = sh.a.huf[(uint32_t)(rb.buf >> (64 - mb))];
        const uint32_t nb = (e >> 8) & 15u;
        if (n < cap) out[n] = (uint8_t)e;
        n++;
        rb.buf <<= nb; rb.avail -= (int32_t)nb; rb.remaining -= (int32_t)nb;
    }
    *count = n; *flags = (rb.remaining != 0) ? 2u : 0u;                 /* :234-241 */
}

#endif /* !CZ_EXEC_ONLY */
/* ------------------------------------------------------------------ frame / block headers */
/* read_frame_header + window_size (frame.cairo:152-284, :106-129).  Lane 0. */
__device__ static __attribute__((noinline)) int cz_parse_frame_header(cz_gcptr p, uint64_t len, CzBroadcast& bc) {
    if (len < 4) return CZ_E_FH_MAGIC_READ;
    const uint32_t magic = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    uint32_t i = 4;
    if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {
        if (len < 8) return CZ_E_FH_DESCRIPTOR_READ;
        bc.d0 = magic; bc.d1 = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
        return CZ_E_FH_SKIP_FRAME;
    }
    if (magic != 0xFD2FB528u) { bc.d0 = magic; return CZ_E_FH_BAD_MAGIC; }
    if (len < i + 1) return CZ_E_FH_DESCRIPTOR_READ;
    const uint32_t d = p[i++];
    const uint32_t single = (d >> 5) & 1;
    uint32_t wd = 0;
    if (!single) { if (len < i + 1) return CZ_E_FH_WINDOW_DESC_READ; wd = p[i++]; }
    const uint32_t didf = d & 3, dl = didf == 3 ? 4 : didf;
    if (dl) { if (len < i + dl) return CZ_E_FH_DICT_ID_READ; i += dl; }
    const uint32_t flag = d >> 6, fl = flag == 0 ? (single ? 1u : 0u) : flag == 1 ? 2u : flag == 2 ? 4u : 8u;
    uint64_t fcs = 0;
    if (fl) {
        if (len < i + fl) return CZ_E_FH_DICT_ID_READ;                  /* frame.cairo:245-270 quirk */
        for (uint32_t k = 0; k < fl; k++) fcs |= (uint64_t)p[i + k] << (8 * k);
        i += fl; if (fl == 2) fcs += 256;
    }
    uint64_t ws;
    if (single) ws = fcs;
    else {
        const uint64_t base = 1ull << (10 + (wd >> 3)); ws = base + (base / 8) * (wd & 7);
        if (ws < 1024) return CZ_E_WINDOW_TOO_SMALL;
        if (ws >= 4123168604160ull) return CZ_E_WINDOW_TOO_BIG;
    }
    bc.hdr_len = i; bc.window_size = ws; bc.has_checksum = (d >> 2) & 1;
    return 0;
}

/* ------------------------------------------------------------------ literals */
struct CzLit { cz_gcptr p; uint32_t len; uint32_t rle; uint8_t byte; };   /* rle=1: `len` copies of byte */

__device__ static inline void cz_lit_coop_copy(uint8_t* dst, const CzLit& lit, uint32_t from, uint32_t n) {
    if (lit.rle) cz_coop_fill(dst, lit.byte, n); else cz_coop_copy(dst, lit.p + from, n);
}

/* LiteralsSection::parse_from_header (literals_section.cairo:81-175) + the serial parts of
 * decompress_literals (literals_section_decoder.cairo:58-117) + SequencesHeader::parse_from_header
 * (sequence_section.cairo:77-114).  Lane 0; results in sh.bc. */
__device__ static __attribute__((noinline)) int cz_parse_sections(cz_gcptr blk, uint32_t bsize, uint32_t stage_hi, int have_literals = 0, int phase = 0) {
    CzBroadcast& bc = sh.bc;
    CzFBits fb; fb.g = blk; fb.stage = sh.a.t1.stage; fb.stage_lo = 0; fb.stage_hi = stage_hi; fb.idx = 0; fb.len = bsize;
    if (bsize == 0) return CZ_E_LS_GETBITS;                             /* :84-90 */
    const uint32_t b0 = cz_fb_byte(fb, 0), type = b0 & 3, fmt = (b0 >> 2) & 3;
    uint32_t need = type <= 1 ? ((fmt == 0 || fmt == 2) ? 1u : (fmt == 1 ? 2u : 3u)) : (fmt <= 1 ? 3u : (fmt == 2 ? 4u : 5u));
    if (bsize < need) return CZ_E_LS_NOT_ENOUGH_BYTES;                  /* :100 */
    uint32_t b1 = need > 1 ? cz_fb_byte(fb, 1) : 0, b2 = need > 2 ? cz_fb_byte(fb, 2) : 0,
             b3 = need > 3 ? cz_fb_byte(fb, 3) : 0, b4 = need > 4 ? cz_fb_byte(fb, 4) : 0;
    uint32_t regen, comp = 0, streams = 0;
    if (type <= 1) {
        if (fmt == 0 || fmt == 2) regen = b0 >> 3; else if (fmt == 1) regen = (b0 >> 4) + (b1 << 4); else regen = (b0 >> 4) + (b1 << 4) + (b2 << 12);
    } else {
        streams = fmt == 0 ? 1 : 4;
        if (fmt <= 1) { regen = (b0 >> 4) + ((b1 & 0x3f) << 4); comp = (b1 >> 6) + (b2 << 2); }
        else if (fmt == 2) { regen = (b0 >> 4) + (b1 << 4) + ((b2 & 3) << 12); comp = (b2 >> 2) + (b3 << 6); }
        else { regen = (b0 >> 4) + (b1 << 4) + ((b2 & 0x3f) << 12); comp = (b2 >> 6) + (b3 << 2) + (b4 << 10); }
    }
    const uint32_t upper = type >= 2 ? comp : (type == 1 ? 1u : regen); /* block_decoder.cairo:160-172 */
    if (bsize - need < upper) return CZ_E_MALFORMED_SECTION_HEADER;     /* block_decoder.cairo:174 */
    bc.lit_type = type; bc.regen = regen; bc.nstreams = streams; bc.lit_total = need + upper; bc.huf_fill = 0;
#ifdef CZ_EXEC_ONLY
    if (type >= 2 && !have_literals) return CZX_FALLBACK;
#else
    if (type >= 2 && !have_literals) {                                  /* literals_section_decoder.cairo:64-117 (skipped when the huff0 kernels
                                                                           already decoded this block's literals, tree included) */
        uint32_t off = need, left = comp;
        if (type == 2) {
            uint32_t used, nsym;
            int e = cz_huf_read_and_rank(blk, left, 0, stage_hi, off, &used, &nsym, phase);
            if (e) return e;
            bc.huf_fill = 1; bc.huf_nsym = nsym;
            if (used > left) return CZ_E_BLOCK_TRUNCATED;               /* (panic) slice(bytes_read, len) :89 */
            off += used; left -= used;
        } else if (sh.huf_max_bits == 0) return CZ_E_LIT_UNINIT_HUF_TABLE;           /* :82-86 */
        if (streams == 4) {
            if (left < 6) return CZ_E_LIT_MISSING_JUMP_HEADER;          /* :92 */
            const uint32_t j1 = cz_fb_byte(fb, off) + (cz_fb_byte(fb, off + 1) << 8);
            const uint32_t j2 = j1 + cz_fb_byte(fb, off + 2) + (cz_fb_byte(fb, off + 3) << 8);
            const uint32_t j3 = j2 + cz_fb_byte(fb, off + 4) + (cz_fb_byte(fb, off + 5) << 8);
            off += 6; left -= 6;
            if (left < j3) return CZ_E_LIT_MISSING_BYTES;               /* :101 */
            bc.stream_off[0] = off;      bc.stream_len[0] = j1;
            bc.stream_off[1] = off + j1; bc.stream_len[1] = j2 - j1;
            bc.stream_off[2] = off + j2; bc.stream_len[2] = j3 - j2;
            bc.stream_off[3] = off + j3; bc.stream_len[3] = left - j3;
        } else { bc.stream_off[0] = off; bc.stream_len[0] = left; }
    }
#endif
    /* sequences header (read early; its error is only reported after the literals decoded) */
    const uint32_t so = need + upper, sl = bsize - so;
    bc.seq_hdr_err = 0; bc.nseq = 0; bc.seq_modes = 0; bc.seq_body_off = so;
    if (sl == 0) bc.seq_hdr_err = CZ_E_SH_NOT_ENOUGH_BYTES;             /* sequence_section.cairo:81 */
    else {
        const uint32_t s0 = cz_fb_byte(fb, so);
        if (s0 == 0) bc.seq_body_off = so + 1;                          /* :85-87 */
        else {
            uint32_t n = 0, hb = 0;
            if (s0 <= 127) { if (sl < 2) bc.seq_hdr_err = CZ_E_SH_NOT_ENOUGH_BYTES; else { n = s0; hb = 1; } }
            else if (s0 <= 254) { if (sl < 3) bc.seq_hdr_err = CZ_E_SH_NOT_ENOUGH_BYTES; else { n = ((s0 - 128) << 8) + cz_fb_byte(fb, so + 1); hb = 2; } }
            else { if (sl < 4) bc.seq_hdr_err = CZ_E_SH_NOT_ENOUGH_BYTES; else { n = cz_fb_byte(fb, so + 1) + (cz_fb_byte(fb, so + 2) << 8) + 0x7F00u; hb = 3; } }
            if (!bc.seq_hdr_err) { bc.nseq = n; bc.seq_modes = cz_fb_byte(fb, so + hb); bc.seq_body_off = so + hb + 1; }
        }
    }
    return 0;
}

__device__ static inline void cz_init_llml() {
    for (uint32_t i = (uint32_t)LANE; i < 36; i += 64) sh.b.c.llml[i] = CZ_LL_BASE[i] | ((uint32_t)CZ_LL_BITS[i] << 24);
    for (uint32_t i = (uint32_t)LANE; i < 53; i += 64) sh.b.c.llml[40 + i] = CZ_ML_BASE[i] | ((uint32_t)CZ_ML_BITS[i] << 24);
}

/* ---- copies of the general path: every global load of a lane is issued before its first store, so a run costs one
 * memory round trip instead of one per byte (a byte loop whose store may alias the next load cannot be pipelined). */
/* the low m (<= 16) bytes of v to d, any alignment, in at most four stores */
__device__ static inline void cz_store_upto16(cz_gptr d, uint4 v, uint32_t m) {
    if (m >= 16) { __builtin_memcpy(d, &v, 16); return; }
    if (m & 8u) { const uint64_t t = ((uint64_t)v.y << 32) | v.x; __builtin_memcpy(d, &t, 8); d += 8; v.x = v.z; v.y = v.w; }
    if (m & 4u) { __builtin_memcpy(d, &v.x, 4); d += 4; v.x = v.y; }
    if (m & 2u) { const uint16_t h = (uint16_t)v.x; __builtin_memcpy(d, &h, 2); d += 2; v.x >>= 16; }
    if (m & 1u) *d = (uint8_t)v.x;
}
#ifndef CZ_EXEC_ONLY
/* ---- self-synchronising parallel huff0 decode ---------------------------------------------
 * A huff0 stream is a prefix code read in one direction, and prefix codes resynchronise: a
 * decoder started in the middle of a codeword falls back onto true codeword boundaries after a
 * few symbols.  Each stream is cut into up to 16 bit ranges, one lane per range (4 streams x 16
 * = the whole wave instead of 4 lanes):
 *   1. every lane decodes its range speculatively from the range boundary, without output, and
 *      notes where it first crosses into the next range (end position) and how many symbols it saw;
 *   2. the true start of range i+1 is the end position of range i: lanes whose start was wrong
 *      redo their range from the corrected start; repeated until no start changes (lane 0 of a
 *      stream starts at the true start, so this is exact after at most 16 rounds; ranges of
 *      >= 256 bits make it 1 round in practice);
 *   3. a segmented prefix sum of the symbol counts gives every range its output offset, and a
 *      last pass decodes and writes.
 * Same results as decoding the stream in one go (literals_section_decoder.cairo:183-243): the
 * count, the exact-end test of the last range and ExtraPadding are reported as before.
 * Every lane reads its own part of the bitstream straight from global memory, 16 bytes at a
 * time, with the next 16 bytes always in flight. */
/* Per-lane reader: a 32-byte register window reloaded by ALL lanes at the same loop iteration
 * (one overlapped HBM/L2 round trip per CZ_GB_SYMS symbols for the whole wave; per-lane refills at
 * data-dependent iterations would stall the wave on nearly every symbol). */
#define CZ_GB_SYMS 20u         /* 20 symbols x 11 bits + 11 bits of lookahead <= 256 - 7 */
struct CzGBits {
    uintptr_t S, E;            /* stream bytes [S, E) */
    uintptr_t LB;              /* lowest address that is safe to read (start of the block): bytes in [LB, S) are
                                  loaded with the stream and masked to zero instead of being fetched one by one */
    int32_t p;                 /* bits of the stream still unread (may go <= 0: zero-extension) */
};
__device__ static inline uint32_t cz_mask_low_bytes(uint32_t w, uint32_t word_lo, uint32_t nbytes) {
    return nbytes >= word_lo + 4 ? 0u : (nbytes > word_lo ? w & (0xFFFFFFFFu << (8 * (nbytes - word_lo))) : w);
}
__device__ static inline uint4 cz_gb_load(uintptr_t a, uintptr_t S, uintptr_t E, uintptr_t LB) {
    uint4 v;
    if (a >= LB && a + 16 <= E) {
        __builtin_memcpy(&v, (CZ_GLOBAL const void*)a, 16);
        if (a < S) {                                                    /* zero the bytes below the stream start */
            const uint32_t nb = (uint32_t)(S - a) > 16 ? 16u : (uint32_t)(S - a);
            v.x = cz_mask_low_bytes(v.x, 0, nb); v.y = cz_mask_low_bytes(v.y, 4, nb); v.z = cz_mask_low_bytes(v.z, 8, nb); v.w = cz_mask_low_bytes(v.w, 12, nb);
        }
        return v;
    }
    uint32_t w[4] = {0, 0, 0, 0};
    if (a + 16 > S && a < E)
        for (uint32_t b = 0; b < 16; b++) { const uintptr_t q = a + b; if (q >= S && q < E) w[b >> 2] |= (uint32_t)(*(cz_gcptr)q) << (8 * (b & 3)); }
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
}
/* One interval: reload the window at the current position, then decode up to CZ_GB_SYMS symbols
 * while p > stop.  `run` = this lane still has work.  Returns symbols decoded. */
__device__ static inline uint32_t cz_gb_interval(CzGBits& g, uint32_t mb, int32_t stop, int run, cz_gptr out, uint32_t cap, uint32_t n0,
                                               const int multi, int* full) {
    uint32_t n = 0;
    *full = 0;
    if (run) {
        const int32_t p = g.p;
        const uintptr_t top = g.S + (uintptr_t)((p + 7) >> 3);          /* exclusive end of the byte holding bit p-1 */
        const uint32_t drop = (uint32_t)((8 - (p & 7)) & 7);            /* bits of that byte above bit p-1 */
        const uint4 hi4 = cz_gb_load(top - 16, g.S, g.E, g.LB), lo4 = cz_gb_load(top - 32, g.S, g.E, g.LB);
        uint64_t buf = ((((uint64_t)hi4.w) << 32) | hi4.z) << drop; int32_t avail = 64 - (int32_t)drop;
        uint32_t q5 = hi4.y, q4 = hi4.x, q3 = lo4.w, q2 = lo4.z, q1 = lo4.y, q0 = lo4.x;   /* unread words, q5 next */
        /* CZ_GB_SYMS symbols in five groups of four, collected in registers: a full interval is
           written with one 16-byte and one 4-byte store (unaligned), a partial one byte by byte */
        uint32_t word[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t grp = 0; grp < CZ_GB_SYMS / 4; grp++) {
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                /* Branch-free: every lane runs every step and a lane that is past its range consumes zero bits.  (Predicated
                   steps made the compiler carry word[] through a chain of v_mov copies and an exec save/restore per symbol.)
                   Refill once per PAIR of symbols (a pair takes at most 22 of the >= 33 bits). */
                if ((j & 1u) == 0) {
                    const int rf = avail <= 32;
                    const uint32_t sh_ = rf ? (uint32_t)(32 - avail) : 0u;
                    buf |= (uint64_t)(rf ? q5 : 0u) << sh_;
                    avail += rf ? 32 : 0;
                    q5 = rf ? q4 : q5; q4 = rf ? q3 : q4; q3 = rf ? q2 : q3; q2 = rf ? q1 : q2; q1 = rf ? q0 : q1; q0 = rf ? 0u : q0;
                }
                const int live = g.p > stop;
                const uint32_t e = sh.a.huf[(uint32_t)(buf >> (64 - mb))];
                uint32_t nb;
                if (multi) {
                    /* counting only: two symbols when the index bits hold both, unless the second could begin at or below
                       `stop` (then one, so that the range ends exactly where it would otherwise) */
                    const uint32_t l2 = g.p - stop > (int32_t)mb ? e >> 12 : 0u;
                    nb = live ? ((e >> 8) & 15u) + l2 : 0u;
                    n += live ? (l2 ? 2u : 1u) : 0u;
                } else {
                    nb = live ? (e >> 8) & 15u : 0u;
                    word[grp] |= (live ? e & 0xFFu : 0u) << (8 * j);
                    n += live ? 1u : 0u;
                }
                buf <<= nb; avail -= (int32_t)nb; g.p -= (int32_t)nb;
                if (grp == CZ_GB_SYMS / 4 - 1 && j == 3) *full = live;  /* the cursor only moves down: every step of the interval ran */
            }
        }
        if (out) {
            uint4 v; v.x = word[0]; v.y = word[1]; v.z = word[2]; v.w = word[3];
            if (n == CZ_GB_SYMS && n0 + CZ_GB_SYMS <= cap) { __builtin_memcpy(out + n0, &v, 16); __builtin_memcpy(out + n0 + 16, &word[4], 4); }
            else {
                /* the last interval of a range (or a stream that runs over its capacity): exactly the bytes that are there,
                   in at most seven stores (lanes end their ranges at different iterations, so this runs often) */
                const uint32_t room = n0 < cap ? cap - n0 : 0u, m = n < room ? n : room;
                (cz_store_upto16)(out + n0, v, m < 16u ? m : 16u);
#pragma unroll
                for (uint32_t t = 16; t < CZ_GB_SYMS - 1; t++) if (t < m) out[n0 + t] = (uint8_t)(word[4] >> (8 * (t & 3)));
            }
        }
    }
    return n;
}
/* decode from position g.p while p > stop (all 64 lanes call this together; `live` lanes work) */
__device__ static inline uint32_t cz_gb_decode(CzGBits& g, uint32_t mb, int32_t stop, int live, cz_gptr out, uint32_t cap, const int multi,
                                            int32_t* ck = nullptr, uint32_t* ckn = nullptr) {
    uint32_t n = 0, it = 0;
    for (;;) {
        const int run = live && g.p > stop;
        if (!__ballot(run)) break;
        int full;
        const uint32_t got = cz_gb_interval(g, mb, stop, run, out, cap, n, multi, &full);
        n += got; it++;
        /* checkpoints of the speculative pass: position and symbol count after 1, 2, 4 and 8 FULL intervals */
        if (ck && run && full) {
            if (it == 1) { ck[0] = g.p; ckn[0] = n; } else if (it == 2) { ck[1] = g.p; ckn[1] = n; }
            else if (it == 4) { ck[2] = g.p; ckn[2] = n; } else if (it == 8) { ck[3] = g.p; ckn[3] = n; }
        }
    }
    return n;
}

/* All huff0 streams of a block; same contract as the sequential decoder: out_k = target + k*seg,
 * at most cap_k bytes written, bc.st_count[k] / bc.st_flags[k] (1 ExtraPadding, 2 stream did not
 * end exactly, 4 count != cap_k). */
__device__ static void cz_huf_streams_par(cz_gcptr blk, cz_gptr target, uint32_t nstreams, uint32_t seg, uint32_t cap_last, int fits) {
    CzBroadcast& bc = sh.bc;
    const uint32_t mb = cz_uni(sh.huf_max_bits);
    const uint32_t k = nstreams == 4 ? (uint32_t)LANE >> 4 : 0, i = (uint32_t)LANE & 15;   /* stream, range */
    const int mine = nstreams == 4 || LANE < 16;
    const uint8_t* S = blk + bc.stream_off[k]; const uint32_t len = bc.stream_len[k]; const uint8_t* E = S + len;
    const uint32_t cap = fits ? (k < 3 && nstreams == 4 ? seg : cap_last) : 0;
    const uint32_t lastb = len ? E[-1] : 0;
    const int padbad = lastb == 0;                                      /* > 8 padding reads: ExtraPadding (:190-207) */
    const int32_t P0 = padbad ? 0 : (int32_t)len * 8 - (int32_t)(__clz((int)lastb) - 24 + 1);
    /* ranges of >= 256 bits, at most 16 */
    uint32_t m = (uint32_t)P0 >> 8; m = m < 1 ? 1 : (m > 16 ? 16 : m);
    const int32_t C = (P0 + (int32_t)m - 1) / (int32_t)m;
    const int live = mine && !padbad && i < m && P0 > 0;
    const int32_t top_b = P0 - (int32_t)i * C;                          /* nominal start of my range */
    const int32_t stop = (i + 1 == m) ? 0 : P0 - (int32_t)(i + 1) * C;  /* decode while p > stop */
    CzGBits g; g.S = (uintptr_t)S; g.E = (uintptr_t)E; g.LB = (uintptr_t)blk; g.p = top_b;
    int32_t s = top_b, e = stop; uint32_t n = 0;
    CZ_PROF_DECL; CZ_PROF_T0();
    const int32_t CK_NONE = (int32_t)0x80000000;
    int32_t ck[4] = { CK_NONE, CK_NONE, CK_NONE, CK_NONE };            /* positions the speculative pass went through */
    uint32_t ckn[4] = { 0, 0, 0, 0 };                                   /* ... and the symbols it had counted there */
    /* the passes that only count step two symbols at a time where the table says so (cz_huf_fill_multi) */
#ifdef CZ_EXP_NO_MULTI
#define CZ_MULTI 0
#else
#define CZ_MULTI 1
#endif
    n = cz_gb_decode(g, mb, stop, live, nullptr, 0, CZ_MULTI, ck, ckn);   /* 1. speculative pass */
    if (live) e = g.p;
    CZ_PROF_ACC(CZ_P_HUF_SPEC);
    /* 2. fix the starts until nothing moves */
    for (int round = 0; round < 17; round++) {
        const int32_t pe = __shfl_up(e, 1u);
        const int changed = live && i > 0 && pe != s;
        if (!__ballot(changed)) break;
        /* A corrected lane rarely has to redo its whole range: as soon as it lands exactly on a
           position its speculative pass went through, the rest of that pass (end position, symbol
           count) is already the truth.  Checkpoint j was taken after 1 << j full intervals, ckn[j] symbols. */
        if (changed) { s = pe; g.p = s; }
        uint32_t nred = 0; int state = changed ? 0 : 2;                 /* 0 redoing, 1 merged into the old trajectory, 2 done */
        for (int j = 0; j < 4; j++) {
            const int try_ck = state == 0 && ck[j] != CK_NONE && ck[j] > stop;
            nred += cz_gb_decode(g, mb, try_ck ? ck[j] : stop, try_ck, nullptr, 0, CZ_MULTI);
            if (try_ck) {
                if (g.p == ck[j]) { n = nred + (n - ckn[j]); state = 1; }   /* e stays */
                else if (g.p <= stop) { n = nred; e = g.p; state = 2; ck[0] = ck[1] = ck[2] = ck[3] = CK_NONE; }
            }
        }
        nred += cz_gb_decode(g, mb, stop, state == 0, nullptr, 0, CZ_MULTI);
        if (state == 0) { n = nred; e = g.p; }
        if (changed) ck[0] = ck[1] = ck[2] = ck[3] = CK_NONE;           /* counts no longer line up with the checkpoints */
    }
    CZ_PROF_ACC(CZ_P_HUF_SYNC);
    /* 3. output offsets (segmented scan over the 16 lanes of a stream) and the writing pass */
    uint32_t incl = live ? n : 0;
    for (int d = 1; d < 16; d <<= 1) { const uint32_t t = __shfl_up(incl, (unsigned)d); if ((int)i >= d) incl += t; }
    const uint32_t total = __shfl(incl, (int)((LANE & ~15) | 15)), off = incl - (live ? n : 0);
    const int32_t e_last = __shfl(e, (int)((LANE & ~15) + (m - 1)));
    {
        g.p = s;
        const uint32_t room = off < cap ? cap - off : 0;
        cz_gb_decode(g, mb, stop, live, target + (uint64_t)k * seg + off, room, 0);
    }
    CZ_PROF_ACC(CZ_P_HUF_WRITE);
    if (mine && i == 0) {
        uint32_t fl = padbad ? 1u : 0u;
        if (!padbad && e_last != 0) fl |= 2u;                           /* :234-241 */
        const uint32_t cnt = padbad ? 0 : total;
        bc.st_count[k] = cnt; bc.st_flags[k] = fl | ((cnt != cap) ? 4u : 0u);
    }
    __syncthreads();
}

/* Huffman literal streams -> `target` (regen bytes).  All lanes enter; returns status
 * (uniform).  literals_section_decoder.cairo:91-178. */
__device__ static __attribute__((noinline)) int cz_decode_huf_literals(cz_gcptr blk, cz_gptr target) {
    CzBroadcast& bc = sh.bc;
    const uint32_t regen = cz_uni(bc.regen), streams = cz_uni(bc.nstreams);
    if (streams == 4) {
        const uint32_t seg = (regen + 3) >> 2;
        const int fits = 3 * seg <= regen;
        cz_huf_streams_par(blk, target, 4, seg, fits ? regen - 3 * seg : 0, fits);
        const uint32_t f0 = cz_uni(bc.st_flags[0]), f1 = cz_uni(bc.st_flags[1]), f2 = cz_uni(bc.st_flags[2]), f3 = cz_uni(bc.st_flags[3]);
        const uint32_t total = cz_uni(bc.st_count[0] + bc.st_count[1] + bc.st_count[2] + bc.st_count[3]);
        __syncthreads();
        /* first failing stream in stream order decides (reference decodes them sequentially) */
        const uint32_t fl[4] = { f0, f1, f2, f3 };
        for (int k = 0; k < 4; k++) {
            if (fl[k] & 1u) return CZ_E_LIT_EXTRA_PADDING;
            if (fl[k] & 2u) return CZ_E_LIT_BITSTREAM_MISMATCH;
        }
        if (total != regen) return CZ_E_LIT_COUNT_MISMATCH;             /* :172 */
        if ((f0 | f1 | f2 | f3) & 4u) {
            /* Valid streams whose symbol counts are not the ceil(regen/4) split (the reference
               concatenates whatever each stream yields, :112-115): redo them back to back. */
            __syncthreads();
            if (LANE == 0) {
                uint32_t at = 0;
                for (int k = 0; k < 4; k++) { uint32_t c, f; cz_huf_stream(blk + bc.stream_off[k], bc.stream_len[k], target + at, regen - at, &c, &f); at += c; }
            }
            __syncthreads();
        }
        return 0;
    }
    cz_huf_streams_par(blk, target, 1, 0, regen, 1);               /* :118-170, no end-of-stream test */
    const uint32_t sf = cz_uni(bc.st_flags[0]), sc = cz_uni(bc.st_count[0]);
    __syncthreads();
    if (sf & 1u) return CZ_E_LIT_EXTRA_PADDING;
    if (sc != regen) return CZ_E_LIT_COUNT_MISMATCH;
    return 0;
}

/* ------------------------------------------------------------------ sequences */
/* maybe_update_fse_tables, serial part (sequence_section_decoder.cairo:405-647): modes,
 * RLE bytes, probability descriptions.  Lane 0.  Sets build_mask / nprobs / acc_log. */
__device__ static __attribute__((noinline)) int cz_parse_seq_tables(cz_gcptr blk, uint32_t bsize, uint32_t stage_lo, uint32_t stage_hi) {
    CzBroadcast& bc = sh.bc;
    uint32_t off = bc.seq_body_off;
    bc.build_mask = 0;
    const uint32_t seq_modes = bc.seq_modes;
    for (int t = 0; t < 3; t++) {                                       /* LL, OF, ML (no local arrays: they would live in scratch memory) */
        const uint32_t left = bsize - off, mode = (seq_modes >> (6 - 2 * t)) & 3, max_log = t == 1 ? 8u : 9u;
        const int miss = t == 0 ? CZ_E_SEQ_MISSING_RLE_BYTE_LL : (t == 1 ? CZ_E_SEQ_MISSING_RLE_BYTE_OF : CZ_E_SEQ_MISSING_RLE_BYTE_ML);
        if (mode == 0) {                                            /* Predefined */
            const int8_t* d = t == 0 ? CZ_LL_DEFAULT : t == 1 ? CZ_OF_DEFAULT : CZ_ML_DEFAULT;
            const uint32_t n = t == 0 ? 36u : t == 1 ? 29u : 53u;
            for (uint32_t s = 0; s < n; s++) sh.a.t3.probs[t][s] = d[s];
            bc.nprobs[t] = n; bc.acc_log[t] = t == 1 ? 5u : 6u; bc.build_mask |= 1u << t;
            sh.fse_rle[t] = -1;
        } else if (mode == 1) {                                         /* RLE */
            if (left == 0) return miss;
            CzFBits fb; fb.g = blk; fb.stage = sh.a.t3.stage; fb.stage_lo = stage_lo; fb.stage_hi = stage_hi; fb.idx = 0; fb.len = bsize;
            sh.fse_rle[t] = (int32_t)cz_fb_byte(fb, off); off += 1;
        } else if (mode == 2) {                                         /* FSE_Compressed */
            CzFBits br; br.g = blk + off; br.len = left; br.idx = 0; br.stage = sh.a.t3.stage; br.stage_lo = 0; br.stage_hi = 0;
            if (off >= stage_lo && off < stage_hi) { br.stage = sh.a.t3.stage + (off - stage_lo); br.stage_hi = stage_hi - off; }
            uint32_t np, lg, used;
            int e = cz_fse_read_probs(br, max_log, sh.a.t3.probs[t], &np, &lg, &used, 100);
            if (e) return e;
            bc.nprobs[t] = np; bc.acc_log[t] = lg; bc.build_mask |= 1u << t;
            sh.fse_rle[t] = -1;
            off += used;
            if (off > bsize) return CZ_E_BLOCK_TRUNCATED;
        }                                                               /* Repeat: keep */
    }
    bc.bitstream_off = off;
    return 0;
}

#endif /* !CZ_EXEC_ONLY */
/* sequence_execution.cairo:85-129; returns the actual offset (0 = ZeroOffset) */
__device__ static inline uint32_t cz_offset_history(uint32_t ov, uint32_t ll, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
    uint32_t a;
    if (ll > 0) a = ov == 1 ? h0 : ov == 2 ? h1 : ov == 3 ? h2 : ov - 3;
    else        a = ov == 1 ? h1 : ov == 2 ? h2 : ov == 3 ? h0 - 1 : ov - 3;
    if (a == 0) return 0;
    if (ll > 0) {
        if (ov == 1) { }
        else if (ov == 2) { h1 = h0; h0 = a; }
        else { h2 = h1; h1 = h0; h0 = a; }
    } else {
        if (ov == 1) { h1 = h0; h0 = a; }
        else { h2 = h1; h1 = h0; h0 = a; }
    }
    return a;
}

struct CzExecCtx {
    cz_gptr out;             /* frame output base */
    uint64_t cap;            /* capacity of the frame output */
    uint64_t produced;       /* bytes already produced in this frame (total_output_counter) */
    uint64_t drained;        /* bytes drained by the host (buffer.len = produced - drained) */
    uint64_t window;
    uint32_t lit_used;
};
/* DecodeBuffer.dict_content (logically just before the first resident byte, position `drained`) and the lag of the reference's
   total_output_counter (cz_device_frame_state.dict_lag) live in LDS */
__device__ static inline uint64_t cz_dict_len() { return ((uint64_t)sh.dict_len[1] << 32) | sh.dict_len[0]; }
__device__ static inline cz_gcptr cz_dict_ptr() { return (cz_gcptr)(uintptr_t)(((uint64_t)sh.dict_ptr[1] << 32) | sh.dict_ptr[0]); }
__device__ static inline uint64_t cz_dict_lag() { return ((uint64_t)sh.dict_lag[1] << 32) | sh.dict_lag[0]; }

/* 16 bytes at s when the whole load lies inside the buffer (`whole`), else the first m bytes one by one */
__device__ static inline uint4 cz_load_upto16(cz_gcptr s, uint32_t m, int whole) {
    uint4 v;
    if (whole) { __builtin_memcpy(&v, s, 16); return v; }
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t b = 0; b < 16; b++) if (b < m) w[b >> 2] |= (uint32_t)s[b] << (8 * (b & 3));
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
}
/* n <= 32 bytes, one lane; wholeA / wholeB: the 16-byte loads at s / s + 16 stay inside the buffer and read nothing this
   copy writes */
__device__ static inline void cz_lane_copy32(cz_gptr d, cz_gcptr s, uint32_t n, int wholeA, int wholeB) {
    const uint32_t nb = n > 16 ? n - 16 : 0;
    const uint4 a = cz_load_upto16(s, n, wholeA);
    uint4 b = uint4{0, 0, 0, 0};
    if (nb) b = cz_load_upto16(s + 16, nb, wholeB);
    (cz_store_upto16)(d, a, n);
    if (nb) (cz_store_upto16)(d + 16, b, nb);
}
/* Up to four runs of at most CZ_QCOPY_MAX bytes at once, sixteen lanes x 16 bytes per step each: run g (source s, n bytes;
   n == 0: none) belongs to lanes 16g..16g+15.  src_lim: no 16-byte load may reach beyond it (the tail is then read byte
   by byte).  Source and destination of a run do not overlap. */
#define CZ_QCOPY_MAX 1024u
__device__ static inline void cz_quarter_copy(cz_gptr d, cz_gcptr s, uint32_t n, cz_gcptr src_lim) {
    const uint32_t l16 = (uint32_t)LANE & 15u;
#pragma unroll 1
    for (uint32_t pos = 16u * l16; __ballot(pos < n); pos += 256u) {
        if (pos < n) {
            const uint32_t m = n - pos < 16u ? n - pos : 16u;
            const uint4 v = cz_load_upto16(s + pos, m, s + pos + 16 <= src_lim);
            (cz_store_upto16)(d + pos, v, m);
        }
    }
}

/* Execution of up to 64 decoded sequences, one per lane (ll, ml, off = resolved offset), in two
 * stages so that the record-driven path can plan one chunk ahead of the data movement:
 *   cz_chunk_plan  output/literal positions by wave scans + every check of
 *                  sequence_execution.cairo:12-66 / decode_buffer.cairo:62-133 (no memory traffic)
 *   cz_chunk_copy  the literal and match copies. */
struct CzPlan { uint32_t ll, ml, off, orel, lrel, sum_ll, sum_tot; int err; };   /* orel/lrel: output / literal position relative to the chunk start */
__device__ static inline CzPlan cz_chunk_plan(const CzExecCtx& x, uint64_t produced, uint32_t lit_used, const CzLit& lit, uint32_t cnt,
                                              uint32_t ll, uint32_t ml, uint32_t off) {
    const int active = (uint32_t)LANE < cnt;
    if (!active) { ll = 0; ml = 0; off = 1; }
    const uint32_t incl_ll = cz_wave_incl_scan(ll), tot = ll + ml, incl_tot = cz_wave_incl_scan(tot);
    CzPlan p; p.ll = ll; p.ml = ml; p.off = off; p.orel = incl_tot - tot; p.lrel = incl_ll - ll;
    p.sum_ll = cz_readlane(incl_ll, 63); p.sum_tot = cz_readlane(incl_tot, 63);
    const uint64_t dst = produced + p.orel + ll;                        /* where the match goes */
    p.err = 0;
    if (x.cap < 0xC0000000ull) {
        /* everything fits 32 bits: one cheap any-lane-in-trouble test; the exact first error and its
           reference code are only worked out below when it fires (inactive lanes and empty literal
           runs can only trip it when it fires anyway or in streaming mode, where they cost the slow path) */
        const uint32_t d32 = (uint32_t)produced + p.orel + ll;
        const int bad = (lit_used + p.lrel + ll > lit.len) | (off == 0) | (off > d32 - (uint32_t)x.drained) | (d32 + ml > (uint32_t)x.cap);
        if (!__ballot(bad)) return p;
    }
    int e = 0;
    if (active) {
        if (ll > 0 && (uint64_t)lit_used + p.lrel + ll > lit.len) e = CZ_E_EXEC_NOT_ENOUGH_LITERALS;   /* :28-36 */
        else if (off == 0) e = CZ_E_EXEC_ZERO_OFFSET;                                               /* :47 */
    }
    /* matches that begin before the first resident byte: dictionary content, if the frame has one and the reference's
       total_output_counter (which skips matches copied wholly from the dictionary: lag) is still inside the window
       (decode_buffer.cairo:65-93) */
    const int reach = active && !e && ml > 0 && (uint64_t)off > dst - x.drained;
    int dictm = 0;
    if (__ballot(reach)) {
        const uint64_t bfd = reach ? (uint64_t)off - (dst - x.drained) : 0;                         /* bytes_from_dict :67 */
        const int cand = reach && bfd <= cz_dict_len();
        const uint32_t lag_lane = cand && bfd >= ml ? ml : 0u;                                      /* :85-90 */
        const uint64_t lag_before = cz_dict_lag() + (cz_wave_incl_scan(lag_lane) - lag_lane);
        if (reach) {
            if (dst - lag_before > x.window) e = CZ_E_EXEC_OFFSET_TOO_BIG;                          /* :92 */
            else if (!cand) e = CZ_E_EXEC_NOT_ENOUGH_DICT;                                          /* :69-75 */
            else dictm = 1;
        }
    }
    if (active && !e && dst + ml > x.cap) e = CZ_E_OUTPUT_TOO_SMALL;
    const unsigned long long emask = __ballot(e != 0);
    if (emask) { const int first = __ffsll((long long)emask) - 1; p.err = cz_unii(__shfl(e, first)); }
    else if (__ballot(dictm)) p.err = -1;                               /* not an error: cz_chunk_copy takes the dictionary path */
    return p;
}
/* up to 16 bytes into the chunk buffer (LDS) at any alignment */
__device__ static inline __attribute__((always_inline)) void cz_lds_store_upto16(uint8_t* d, uint4 v, uint32_t m) {
    if (m >= 16u) { __builtin_memcpy(d, &v, 16); return; }
    if (m & 8u) { const uint64_t t = ((uint64_t)v.y << 32) | v.x; __builtin_memcpy(d, &t, 8); d += 8; v.x = v.z; v.y = v.w; }
    if (m & 4u) { __builtin_memcpy(d, &v.x, 4); d += 4; v.x = v.y; }
    if (m & 2u) { const uint16_t h = (uint16_t)v.x; __builtin_memcpy(d, &h, 2); d += 2; v.x >>= 16; }
    if (m & 1u) *d = (uint8_t)v.x;
}
/* First NL literal bytes and first NMB match bytes (plain far matches) of every lane into the chunk
 * buffer: all global loads first, then byte writes whose address is the lane's dump byte when the
 * lane has fewer bytes (an address select is cheaper than masking the lane off). */
template <int NL, int NMB>
__device__ static inline void cz_copy_group0(uint8_t* ob, uint32_t orel, uint32_t drel, uint32_t ll, uint32_t ml, const uint8_t* ls, const uint8_t* ms,
                                             const CzLit& lit, int far_plain, int lit_wide) {
    uint32_t lw[2] = {0, 0}, mw[NMB / 4];
    const int lg = ll > 0 && !lit.rle;
    if (lit_wide) {
        if (lg) { if (NL <= 2) { uint16_t h; __builtin_memcpy(&h, ls, 2); lw[0] = h; } else __builtin_memcpy(&lw[0], ls, 4); if (NL > 4) __builtin_memcpy(&lw[1], ls + 4, 4); }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < NL; j++) if (lg && j < ll) lw[j >> 2] |= (uint32_t)ls[j] << (8 * (j & 3));
    }
    /* no overlap: 4-byte loads; the over-read stays below dst + 3 <= dst + ml <= cap */
#pragma unroll
    for (uint32_t j = 0; j < NMB / 4; j++) { mw[j] = 0; if (far_plain && 4 * j < ml) __builtin_memcpy(&mw[j], ms + 4 * j, 4); }
    if (lit.rle) { lw[0] = 0x01010101u * lit.byte; lw[1] = lw[0]; }
    const uint32_t dump = CZ_OBUF_BYTES + 16 + (uint32_t)LANE;
#pragma unroll
    for (uint32_t j = 0; j < NL; j++) ob[j < ll ? orel + j : dump] = (uint8_t)(lw[j >> 2] >> (8 * (j & 3)));
    const uint32_t mlim = far_plain ? ml : 0;
#pragma unroll
    for (uint32_t j = 0; j < NMB; j++) ob[j < mlim ? drel + j : dump] = (uint8_t)(mw[j >> 2] >> (8 * (j & 3)));
}
/* Unaligned accesses to the chunk buffer through LDS-address-space pointers (the conversion from the generic pointer folds away
   once the callers are inlined). */
#ifdef CZ_EMU
#define CZ_LDS_AS
#else
#define CZ_LDS_AS __attribute__((address_space(3)))
#endif
struct __attribute__((packed)) CzP16 { uint16_t v; };
struct __attribute__((packed)) CzP32 { uint32_t v; };
struct __attribute__((packed)) CzP64 { uint64_t v; };
typedef CZ_LDS_AS uint8_t* cz_lptr;
__device__ static inline cz_lptr cz_lds(uint8_t* p) { return (cz_lptr)p; }
__device__ static inline uint64_t cz_lds_r64(cz_lptr p) { return ((CZ_LDS_AS const CzP64*)p)->v; }
/* exactly n <= 8 bytes of v */
__device__ static inline void cz_lds_wn(cz_lptr q, uint64_t v, uint32_t n) {
    if (n >= 8u) { ((CZ_LDS_AS CzP64*)q)->v = v; return; }
    if (n & 4u) { ((CZ_LDS_AS CzP32*)q)->v = (uint32_t)v; q += 4; v >>= 32; }
    if (n & 2u) { ((CZ_LDS_AS CzP16*)q)->v = (uint16_t)v; q += 2; v >>= 16; }
    if (n & 1u) *q = (uint8_t)v;
}
/* n bytes to ob + d from ob + d - off inside the chunk buffer: the forward byte copy of decode_buffer.cairo:95-127, eight bytes a
   step.  With off < n the output is periodic from d - off on, so any earlier multiple of off is as good a distance: the distance
   doubles until it covers a step.  (A step reads at most 7 bytes beyond what it uses.) */
__device__ static inline void cz_lane_l2l(uint8_t* ob_, uint32_t d, uint32_t off, uint32_t n) {
    const cz_lptr ob = cz_lds(ob_);
    uint32_t copied = 0, dist = off;
    while (copied < n) {
        while (dist < 8u && 2u * dist <= off + copied) dist += dist;
        uint32_t step = n - copied < dist ? n - copied : dist;
        if (step > 8u) step = 8u;
        cz_lds_wn(ob + d + copied, cz_lds_r64(ob + d + copied - dist), step);
        copied += step;
    }
}
/* Chunks with longer runs (a literal run above 4 or a match above 8 bytes; none above CZ_OBUF_MAXLEN): what comes from global
 * memory comes in ONE round trip (two for far matches above 64 bytes).  The literals of a chunk are one contiguous stretch of the
 * literal buffer — [lit_used, lit_used + sum_ll) —, so the wave fetches them with 16-byte loads, a KiB per instruction, into a
 * staging area in LDS, and every lane moves its run LDS -> LDS from there; the far matches' bytes are loaded by their lanes, four
 * 16-byte pieces at a time, every load issued before the first use.  (One piece per loop turn, first the literal runs and then
 * the matches, was a round trip per 16 bytes of the longest run of each kind.)  A 16-byte load may read beyond its run where
 * that stays inside the literal buffer / the frame's output buffer; only the run's bytes are written. */
__device__ static inline void cz_copy_long_runs(uint8_t* ob, uint8_t* stg, uint32_t orel, uint32_t drel, uint32_t lrel, uint32_t ll, uint32_t ml, uint32_t sum_ll,
                                                const CzLit& lit, uint32_t lit_used, cz_gcptr ms, uint64_t ms_pos, uint64_t cap, int far_plain) {
    constexpr uint32_t NST = CZ_OBUF_BYTES / 1024u;
    const int staged = !lit.rle;
    uint4 sv[NST], mv[4];
#pragma unroll
    for (uint32_t t = 0; t < NST; t++) {
        const uint32_t lo = 1024u * t + 16u * (uint32_t)LANE;
        sv[t] = uint4{0, 0, 0, 0};
        if (staged && 1024u * t < sum_ll) {                              /* (uniform) */
#if defined(CZ_EXP_LITNEAR)   /* diagnostic only (wrong output): the literals always from the start of the literal buffer (cache hits) */
            if (lo < sum_ll) sv[t] = cz_load_upto16((cz_gcptr)lit.p + lo, sum_ll - lo < 16u ? sum_ll - lo : 16u, (uint64_t)lo + 16u <= lit.len);
#elif defined(CZ_EXP_NOLOADS)  /* diagnostic only (wrong output): no loads at all here */
            if (lo < sum_ll) sv[t] = uint4{lo, lo, lo, lo};
#else
            if (lo < sum_ll) sv[t] = cz_load_upto16((cz_gcptr)lit.p + lit_used + lo, sum_ll - lo < 16u ? sum_ll - lo : 16u, (uint64_t)lit_used + lo + 16u <= lit.len);
#endif
        }
    }
    for (uint32_t k = 0; k == 0 || __ballot(far_plain && k < ml); k += 64) {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t at = k + 16u * j;
            mv[j] = uint4{0, 0, 0, 0};
#ifdef CZ_EXP_NOLOADS
            if (far_plain && at < ml) mv[j] = uint4{ml, ml, ml, ml};
#else
            if (far_plain && at < ml) mv[j] = cz_load_upto16(ms + at, ml - at < 16u ? ml - at : 16u, ms_pos + at + 16u <= cap);
#endif
        }
        if (k == 0) {
#pragma unroll
            for (uint32_t t = 0; t < NST; t++) if (staged && 1024u * t < sum_ll) __builtin_memcpy(stg + 1024u * t + 16u * (uint32_t)LANE, &sv[t], 16);
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint32_t at = k + 16u * j;
            if (far_plain && at < ml) (cz_lds_store_upto16)(ob + drel + at, mv[j], ml - at < 16u ? ml - at : 16u);
        }
    }
    cz_wave_sync();
    for (uint32_t k = 0; __ballot(k < ll); k += 16) {
        if (k < ll) {
            uint4 v;
            if (lit.rle) { const uint32_t w = 0x01010101u * lit.byte; v = uint4{w, w, w, w}; }
#ifdef CZ_EXP_NOSTAGE   /* diagnostic only: the literals by their lanes from global memory, a round trip per 16 bytes (the scheme before) */
            else v = cz_load_upto16((cz_gcptr)lit.p + lit_used + lrel + k, ll - k < 16u ? ll - k : 16u, (uint64_t)lit_used + lrel + k + 16u <= lit.len);
#else
            else __builtin_memcpy(&v, stg + lrel + k, 16);
#endif
            (cz_lds_store_upto16)(ob + orel + k, v, ll - k < 16u ? ll - k : 16u);
        }
    }
}
__device__ static int cz_chunk_copy(CzExecCtx& x, const CzLit& lit, const CzPlan& p) {
    CZ_PROF_DECL; CZ_PROF_T0();
    const uint32_t ll = p.ll, ml = p.ml, off = p.off, tot = ll + ml, incl_tot = p.orel + tot;
    const uint32_t sum_ll = p.sum_ll, sum_tot = p.sum_tot;
    const int active = tot > 0;
    const uint32_t lit_start = x.lit_used + p.lrel;
    const uint64_t out_start = x.produced + (uint64_t)p.orel;
    const uint64_t dst = out_start + ll;                                /* where the match goes */

    /* Fast path for chunks of short sequences: the chunk's output is assembled in LDS and written out
     * with coalesced stores.  Literal bytes and match bytes whose source lies before the chunk are
     * fetched from global memory with every load of a lane issued before its first use (one memory
     * round trip per phase instead of one per byte); matches that read the chunk's own output are
     * resolved by dependency rounds inside LDS. */
    if (p.err == 0 && sum_tot <= CZ_OBUF_BYTES && !__ballot(ll > CZ_OBUF_MAXLEN || ml > CZ_OBUF_MAXLEN)) {
        CZ_PROF_CNT(CZ_P_N_FAST);
        uint8_t* ob = sh.a.t4.obuf;
        const uint32_t orel = incl_tot - tot, drel = orel + ll;
        uint8_t* const cout = x.out + x.produced;                       /* chunk output base */
        /* source range relative to the chunk base: [srel, srel + span) with span = min(off, ml) */
        const int32_t srel = (int32_t)drel - (int32_t)(off < 0x40000000u ? off : 0x40000000u);
        const uint32_t span = off < ml ? off : ml;
        const int far = ml > 0 && srel + (int32_t)span <= 0;            /* every source byte precedes the chunk */
        const int near = ml > 0 && !far;
        const int far_plain = far && off >= ml, far_period = far && off < ml;
        const uint8_t* ls = lit.p + lit_start;
#ifdef CZ_EXP_NEARSRC   /* diagnostic only (wrong output): far match sources read next to the chunk instead of anywhere in the window */
        const uint8_t* ms = x.produced >= 4096 ? cout - 8 - ((drel * 7u + off) & 1023u) : cout + (drel - (uint64_t)off);
#else
        const uint8_t* ms = cout + (drel - (uint64_t)off);
#endif
        /* first group of every lane: all loads, then all LDS writes; sized by the longest run of the chunk */
        const int lit_wide = !__ballot(ll > 0 && !lit.rle && (uint64_t)lit_start + 8 > lit.len);   /* 8-byte literal loads stay inside the buffer */
        const unsigned long long big = __ballot(ll > 4 || ml > 8), mid = __ballot(ll > 2 || ml > 4);
        if (!mid) cz_copy_group0<2, 4>(ob, orel, drel, ll, ml, ls, ms, lit, far_plain, lit_wide);
        else if (!big) cz_copy_group0<4, 8>(ob, orel, drel, ll, ml, ls, ms, lit, far_plain, lit_wide);
        else cz_copy_long_runs(ob, sh.a.t4.lstage, orel, drel, p.lrel, ll, ml, sum_ll, lit, x.lit_used, (cz_gcptr)ms, dst - off, x.cap, far_plain);
        if (__ballot(far_period)) {                                     /* period-off pattern (decode_buffer.cairo:101-120) */
            uint8_t mt[8]; uint32_t idx = 0;
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) if (far_period && j < ml) { mt[j] = ms[idx]; idx = idx + 1 == off ? 0 : idx + 1; }
#pragma unroll
            for (uint32_t j = 0; j < 8; j++) if (far_period && j < ml) ob[drel + j] = mt[j];
            if (far_period) for (uint32_t k = 8; k < ml; k++) { ob[drel + k] = ms[idx]; idx = idx + 1 == off ? 0 : idx + 1; }
        }
        cz_wave_sync();
        CZ_PROF_ACC(CZ_P_LITCOPY);
        /* matches that read this chunk's output: rounds.  W = destination of the first undone match;
           a match may go once its source range (clipped to its own destination) lies below W. */
#ifdef CZ_EXP_NOROUNDS   /* diagnostic only (wrong output): ... without the dependency rounds */
        int done = 1;
#else
        int done = !near;
#endif
        const int32_t send = srel + (int32_t)span;                      /* <= drel */
        for (;;) {
            const unsigned long long pend = __ballot(!done);
            if (!pend) break;
            const int f = __ffsll((long long)pend) - 1;
            const int32_t W = (int32_t)cz_readlane(drel, cz_unii(f));
            const int ready = !done && send <= W;
            if (ready) {
#if defined(CZ_EXEC_ONLY) && !defined(CZ_EXP_L2L_OFF)   /* (cz_decode_frames_kernel keeps the byte loop: with cz_lane_l2l in it this compiler ended
                                                           its build with "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base") */
                if (srel >= 0) cz_lane_l2l(ob, drel, off, ml);          /* the source lies inside the chunk buffer: 8 bytes a step */
                else
#endif
                {
                    uint32_t idx = 0;
                    for (uint32_t k = 0; k < ml; k++) {
                        const int32_t q = srel + (int32_t)idx;
                        ob[drel + k] = q < 0 ? cout[q] : ob[q];
                        idx = idx + 1 == off ? 0 : idx + 1;
                    }
                }
            }
            if (ready) done = 1;
            cz_wave_sync();
        }
        /* write the assembled chunk: 16 bytes per lane per pass (the global address need not be aligned).
           Later loads of these bytes by this wave are ordered behind the stores by the memory pipeline. */
#ifndef CZ_EXP_NOSTORE   /* diagnostic only (wrong output): ... without the stores of the assembled chunk */
#ifdef CZ_EXP_STORE4
        for (uint32_t i = 4u * (uint32_t)LANE; i < sum_tot; i += 256) {
            if (i + 4 <= sum_tot) { uint32_t w = *(const uint32_t*)(ob + i); __builtin_memcpy(cout + i, &w, 4); }
            else for (uint32_t j = i; j < sum_tot; j++) cout[j] = ob[j];
        }
#else
        for (uint32_t i = 16u * (uint32_t)LANE; i < sum_tot; i += 1024) {
            if (i + 16 <= sum_tot) { const uint4 w = *(const uint4*)(ob + i); __builtin_memcpy(cout + i, &w, 16); }
            else for (uint32_t j = i; j < sum_tot; j++) cout[j] = ob[j];
        }
#endif
#endif
        cz_wave_sync();
        CZ_PROF_ACC(CZ_P_MATCH);
        x.produced += sum_tot; x.lit_used += sum_ll;
        return 0;
    }

    /* literals: runs of up to 32 bytes per lane, up to CZ_QCOPY_MAX four at a time by quarter waves, longer ones by the wave */
    CZ_PROF_CNT(CZ_P_N_GENERAL);
    {
        cz_gptr od = (cz_gptr)x.out + out_start; cz_gcptr lsrc = (cz_gcptr)lit.p + lit_start; cz_gcptr llim = (cz_gcptr)lit.p + lit.len;
        if (lit.rle) {
            if (active && ll > 0 && ll <= 32) { const uint32_t w = 0x01010101u * lit.byte; const uint4 v = uint4{w, w, w, w}; (cz_store_upto16)(od, v, ll); if (ll > 16) (cz_store_upto16)(od + 16, v, ll - 16); }
        } else if (active && ll > 0 && ll <= 32) cz_lane_copy32(od, lsrc, ll, lsrc + 16 <= llim, lsrc + 32 <= llim);
        unsigned long long qm = __ballot(ll > 32 && ll <= CZ_QCOPY_MAX && !lit.rle);
        while (qm) {
            int j = -1;                                                 /* the run of my quarter */
#pragma unroll
            for (int g = 0; g < 4; g++) { const int f = qm ? __ffsll((long long)qm) - 1 : -1; if (f >= 0) qm &= qm - 1; if ((LANE >> 4) == g) j = f; }
            const int jj = j < 0 ? 0 : j;
            const uint32_t nj = __shfl(ll, jj), ls = __shfl(lit_start, jj), n = j < 0 ? 0u : nj;   /* every lane takes part in the shuffles */
            const uint64_t os = ((uint64_t)__shfl((uint32_t)(out_start >> 32), jj) << 32) | __shfl((uint32_t)out_start, jj);
            cz_quarter_copy((cz_gptr)x.out + os, (cz_gcptr)lit.p + ls, n, llim);
        }
        unsigned long long lm = __ballot(ll > CZ_QCOPY_MAX || (ll > 32 && lit.rle));
        while (lm) {
            const int j = __ffsll((long long)lm) - 1; lm &= lm - 1;
            const uint32_t n = cz_uni(__shfl(ll, j)), ls = cz_uni(__shfl(lit_start, j));
            const uint64_t os = ((uint64_t)cz_uni(__shfl((uint32_t)(out_start >> 32), j)) << 32) | cz_uni(__shfl((uint32_t)out_start, j));
            cz_lit_coop_copy(x.out + os, lit, ls, n);
        }
    }
    /* No wait for the stores: a wave's vector-memory instructions reach the memory pipeline in program order, so the loads
       below see them (the LDS path above relies on the same; s_waitcnt vmcnt(0) here and after every round was a third
       of this path's time) */
    cz_wave_sync();
    CZ_PROF_ACC(CZ_P_LITCOPY);

    /* matches: dependency rounds.  W = first byte not yet guaranteed written = match
       destination of the first undone sequence; a sequence may go once its source range
       (clipped to its own destination for self-overlap) lies below W. */
    int done = !(active && ml > 0);
    /* dictionary matches (p.err < 0: the plan found some, and no error): a lane whose match begins before the first
       resident byte copies bytes_from_dict bytes from the end of the dictionary content and goes on from the START of the
       resident buffer (decode_buffer.cairo:77-90) — when every earlier sequence is done, one byte at a time */
    const int dictm = p.err < 0 && active && ml > 0 && (uint64_t)off > dst - x.drained;
    if (p.err < 0) {
        const uint64_t bfd = dictm ? (uint64_t)off - (dst - x.drained) : 0;
        const uint32_t lag_lane = dictm && bfd >= ml ? ml : 0u;
        const uint64_t lag = cz_dict_lag() + cz_readlane(cz_wave_incl_scan(lag_lane), 63);
        cz_wave_sync();
        if (LANE == 0) { sh.dict_lag[0] = (uint32_t)lag; sh.dict_lag[1] = (uint32_t)(lag >> 32); }
        cz_wave_sync();
    }
    const uint64_t src = dst - off;
    const uint64_t src_end = (src + ml < dst) ? src + ml : dst;
    for (;;) {
        const unsigned long long pend = __ballot(!done);
        if (!pend) break;
        const int f = __ffsll((long long)pend) - 1;
        const uint64_t W = ((uint64_t)__shfl((uint32_t)(dst >> 32), f) << 32) | __shfl((uint32_t)dst, f);
        if (dictm && !done && LANE == f) {
            const uint64_t bfd = (uint64_t)off - (dst - x.drained);
            cz_gptr d = (cz_gptr)x.out + dst; cz_gcptr dc = cz_dict_ptr() + (cz_dict_len() - bfd); cz_gcptr head = (cz_gcptr)x.out + x.drained;
            for (uint32_t k = 0; k < ml; k++) d[k] = k < bfd ? dc[k] : head[k - bfd];
            done = 1;
        }
        const int ready = !done && !dictm && src_end <= W;
        CZ_PROF_CNT(CZ_P_N_ROUNDS);
        if (ready && ml <= 32) {
            cz_gptr d = (cz_gptr)x.out + dst; cz_gcptr sp = (cz_gcptr)x.out + src;
            /* both 16-byte loads precede the stores: they must not read what this copy writes, and stay below dst (<= cap) */
            if (off >= 32 || (off >= 16 && ml <= 16)) cz_lane_copy32(d, sp, ml, 1, 1);
            else for (uint32_t k = 0; k < ml; k++) d[k] = sp[k];       /* forward byte copy == decode_buffer.cairo:101-120 */
        }
        unsigned long long qm = __ballot(ready && ml > 32 && ml <= CZ_QCOPY_MAX && off >= ml);
        while (qm) {
            int j = -1;
#pragma unroll
            for (int g = 0; g < 4; g++) { const int ff = qm ? __ffsll((long long)qm) - 1 : -1; if (ff >= 0) qm &= qm - 1; if ((LANE >> 4) == g) j = ff; }
            const int jj = j < 0 ? 0 : j;
            const uint32_t nj = __shfl(ml, jj), o = __shfl(off, jj), n = j < 0 ? 0u : nj;
            const uint64_t dj = ((uint64_t)__shfl((uint32_t)(dst >> 32), jj) << 32) | __shfl((uint32_t)dst, jj);
            cz_gptr d = (cz_gptr)x.out + dj;
            cz_quarter_copy(d, d - o, n, d);                            /* loads stay below the run's own destination */
        }
        unsigned long long big = __ballot(ready && ml > 32 && !(ml <= CZ_QCOPY_MAX && off >= ml));
        while (big) {
            const int j = __ffsll((long long)big) - 1; big &= big - 1;
            const uint32_t n = cz_uni(__shfl(ml, j)), o = cz_uni(__shfl(off, j));
            const uint64_t dj = ((uint64_t)cz_uni(__shfl((uint32_t)(dst >> 32), j)) << 32) | cz_uni(__shfl((uint32_t)dst, j));
            uint8_t* d = x.out + dj; const uint8_t* s = d - o;
            CZ_PROF_CNT(CZ_P_N_BIG);
            if (o >= n) cz_coop_copy(d, s, n);
            else {
                /* overlapping (decode_buffer.cairo:101-120 copies `o` bytes at a time): the output is periodic with period o
                   from s on, so any earlier multiple of o is as good a source — the copied length doubles every time
                   (o, 2o, 4o, ...: log2(n / o) non-overlapping copies instead of n / o) */
                uint32_t copied = 0, L = o;
                while (copied < n) {
                    const uint32_t len = n - copied < L ? n - copied : L;
                    cz_coop_copy(d + copied, d + copied - L, len);
                    cz_wave_sync();
                    copied += len; L += L;
                }
            }
        }
        if (ready) done = 1;
        cz_wave_sync();
    }
    CZ_PROF_ACC(CZ_P_MATCH);
    x.produced += sum_tot; x.lit_used += sum_ll;
    return 0;
}

__device__ static int cz_execute_chunk(CzExecCtx& x, const CzLit& lit, uint32_t cnt, uint32_t ll, uint32_t ml, uint32_t off) {
    const CzPlan p = cz_chunk_plan(x, x.produced, x.lit_used, lit, cnt, ll, ml, off);
    if (p.err > 0) return p.err;
    return cz_chunk_copy(x, lit, p);
}

/* ---- sequences bitstream ring (LDS) ------------------------------------------------------
 * The reversed bitstream [S, E) is staged into sh.a.t4.ring by coalesced 16-byte loads, indexed by
 * ABSOLUTE address (ring[a & 2047]) so that global and LDS accesses are both 16-byte aligned.
 * Bytes outside [S, E) are written as zero, which is exactly the reference reader's
 * zero-extension below bit 0 (bit_reader_reverse.cairo:147-159). */
#ifndef CZ_EXEC_ONLY
__device__ static void cz_ring_load_block(const uint8_t* S, const uint8_t* E, uintptr_t block) {
    const uintptr_t a = block + 16u * (uintptr_t)LANE;
    uint4 v;
    if (a >= (uintptr_t)S && a + 16 <= (uintptr_t)E) v = *(CZ_GLOBAL const uint4*)a;
    else {
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t b = 0; b < 16; b++) { const uintptr_t q = a + b; if (q >= (uintptr_t)S && q < (uintptr_t)E) w[b >> 2] |= (uint32_t)(*(cz_gcptr)q) << (8 * (b & 3)); }
        v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    }
    const uint32_t slot = (uint32_t)(a & (CZ_RING_BYTES - 1));
    *(uint4*)&sh.a.t4.ring[slot] = v;
    if (slot == CZ_RING_BYTES - 16) { *(uint32_t*)&sh.a.t4.mirror[8] = v.z; *(uint32_t*)&sh.a.t4.mirror[12] = v.w; }
}
/* 64 stream bits whose most significant bit is stream bit t (t >= 0); lower bits follow */
__device__ static inline uint64_t cz_ring_window(uint32_t sbits, int32_t t) {
    const uint32_t g = sbits + (uint32_t)t, wi = g >> 5, r = (g & 31) + 1;
    const uint32_t* rw = (const uint32_t*)sh.a.t4.ring;
    const uint32_t M = CZ_RING_BYTES / 4 - 1;
    const uint32_t w2 = rw[wi & M], w1 = rw[(wi - 1) & M], w0 = rw[(wi - 2) & M];
    const uint32_t hi = (uint32_t)((((uint64_t)w2 << 32) | w1) >> r), lo = (uint32_t)((((uint64_t)w1 << 32) | w0) >> r);
    return ((uint64_t)hi << 32) | lo;
}
#endif /* !CZ_EXEC_ONLY */
/* n-bit field (n <= 32) starting o bits below the top of W (o + n <= 64) */
__device__ static inline uint32_t cz_field(uint64_t W, uint32_t o, uint32_t n) { return (uint32_t)(((W << o) >> 1) >> (63 - n)); }

/* Repeat-offset history (sequence_execution.cairo:85-129) for up to 64 sequences, one per lane, by a
 * DPP wave scan.  Every transform either permutes the three history slots or pushes an offset in
 * front, so after any prefix of sequences a slot holds either what one of the three slots held at
 * the start of the chunk or the offset pushed by some lane.  A transform is one packed word (see below);
 * composing "P, then Q" is a byte permute with Q's word as the selector.
 * The one transform that is not of this kind, offset_value 3 with no literals,
 * pushes h0 - 1: it scans as a push, and the few such values of a chunk are filled in afterwards in
 * lane order (each needs only slot 0 before its lane).  Advances the uniform history (h0,h1,h2) and
 * returns the lane's actual offset = slot 0 after its own transform. */
#define CZ_T_ID 0x00020100u
__device__ static inline uint32_t cz_pick3(uint32_t k, uint32_t a0, uint32_t a1, uint32_t a2) { const uint32_t pa = (k & 1u) ? a1 : a0; return (k & 2u) ? a2 : pa; }
__device__ static inline uint32_t cz_history(uint32_t cnt, uint32_t ll, uint32_t ov, uint32_t& h0, uint32_t& h1, uint32_t& h2) {
    const int active = (uint32_t)LANE < cnt;
    const int dec = active && ov == 3 && ll == 0;
    /* A transform is ONE packed word T (byte k = slot k): 0,1,2 = the slot holds what slot T_k held before, 0x80 | lane = it
       holds the offset pushed by that lane; M = 0xFF in the bytes of T that are "pushed".  "P, then Q": v_perm_b32 with Q's T
       as the selector picks P's byte for 0..2 and yields the constant 0xFF for a selector byte >= 13 — which is exactly the
       new M for Q's own pushes, and in T those bytes are put back from Q with one v_bfi.  Five instructions per scan step. */
    uint32_t T, M;
    {
        /* 0 keep, 1 swap h0/h1, 2 rotate h2 to the front, 3 push — selected on single bits (no branches) */
        const uint32_t kind = !active ? 0u : (ov > 3 ? 3u : ov - (ll > 0 ? 1u : 0u));
        const uint32_t t01 = (kind & 1u) ? 0x00020001u : CZ_T_ID, t23 = (kind & 1u) ? (0x00010080u | (uint32_t)LANE) : 0x00010002u;
        T = (kind & 2u) ? t23 : t01;
        M = kind == 3u ? 0xFFu : 0u;
    }
#define CZ_HT_STEP(CTRL, RM) do { const uint32_t pT = cz_dpp<CTRL, RM>(CZ_T_ID, T), pM = cz_dpp<CTRL, RM>(0u, M); \
        const uint32_t R = __builtin_amdgcn_perm(T, pT, T); const uint32_t Mn = __builtin_amdgcn_perm(M, pM, T); \
        T = (T & M) | (R & ~M); M = Mn; } while (0)
    CZ_HT_STEP(CZ_DPP_SHR1, 0xF); CZ_HT_STEP(CZ_DPP_SHR2, 0xF); CZ_HT_STEP(CZ_DPP_SHR4, 0xF); CZ_HT_STEP(CZ_DPP_SHR8, 0xF);
    CZ_HT_STEP(CZ_DPP_BCAST15, 0xA); CZ_HT_STEP(CZ_DPP_BCAST31, 0xC);
#undef CZ_HT_STEP
    /* pushed values: ov - 3, or (h0 before the lane) - 1 for the h0 - 1 sequences, resolved in lane order */
    uint32_t pv = ov - 3;
    const unsigned long long dm = __ballot(dec);
    if (dm) {
        const uint32_t eT = cz_dpp<CZ_DPP_WAVE_SHR1, 0xF>(CZ_T_ID, T);      /* transform of everything before the lane */
        for (unsigned long long m = dm; m; m &= m - 1) {
            const int j = cz_unii(__ffsll((long long)m) - 1);
            const uint32_t bT = cz_readlane(eT, j) & 0xFFu;
            const uint32_t r = (bT & 0x80u) ? cz_readlane(pv, cz_unii((int)(bT & 63u))) : cz_pick3(bT & 3u, h0, h1, h2);
            if (LANE == j) pv = r - 1;
        }
    }
    const uint32_t pushed = __shfl(pv, (int)(T & 63u));                 /* every lane takes part */
    const uint32_t actual = (T & 0x80u) ? pushed : cz_pick3(T & 3u, h0, h1, h2);
    const int lastl = cz_unii((int)cnt - 1);
    const uint32_t fT = cz_readlane(T, lastl);
    uint32_t n[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t tk = (fT >> (8 * k)) & 0xFFu;
        const uint32_t pk = cz_readlane(pv, cz_unii((int)(tk & 63u)));
        n[k] = (tk & 0x80u) ? pk : cz_pick3(tk & 3u, h0, h1, h2);
    }
    h0 = cz_uni(n[0]); h1 = cz_uni(n[1]); h2 = cz_uni(n[2]);
    return actual;
}
__device__ static int cz_history_and_execute(CzExecCtx& x, const CzLit& lit, uint32_t cnt, uint32_t ll, uint32_t ml, uint32_t ov,
                                             uint32_t& h0, uint32_t& h1, uint32_t& h2) {
    CZ_PROF_DECL; CZ_PROF_T0();
    const uint32_t actual = cz_history(cnt, ll, ov, h0, h1, h2);
    CZ_PROF_ACC(CZ_P_EXTRACT);
    return cz_execute_chunk(x, lit, cnt, ll, ml, actual);
}

#ifndef CZ_EXEC_ONLY
/* decode_sequences + execute_sequences for one block.  All lanes.
 * sequence_section_decoder.cairo:35-297, sequence_execution.cairo:12-83.
 * Lane 0 runs only the serial core of the three interleaved FSE state machines (one LDS
 * round trip per sequence: three table entries + the bit window) and records (bit position,
 * states) per sequence; the 64 lanes then extract the extra bits, resolve repeat offsets with
 * a wave scan over history transforms and execute their sequence. */
__device__ static int cz_sequences(cz_gcptr blk, uint32_t bsize, CzExecCtx& x, const CzLit& lit) {
    CzBroadcast& bc = sh.bc;
    const uint32_t nseq = cz_uni(bc.nseq);
    const uint8_t* S = blk + cz_uni(bc.bitstream_off); const uint8_t* E = blk + bsize;
    const uint32_t sbits = (uint32_t)((uintptr_t)S & (CZ_RING_BYTES - 1)) * 8u;
    CZ_PROF_DECL; CZ_PROF_T0();
    __syncthreads();
    /* per-table constants: RLE tables behave like a one-entry table (num_bits 0, base 0) */
    const int32_t rLL = cz_unii(sh.fse_rle[0]), rOF = cz_unii(sh.fse_rle[1]), rML = cz_unii(sh.fse_rle[2]);
    const uint32_t fLL = rLL >= 0 ? (CZ_FSE_PACK(rLL, 0, 0) | cz_fse_code_bits(sh.b.c.llml, 0, (uint32_t)rLL)) : 0;
    const uint32_t fOF = rOF >= 0 ? (CZ_FSE_PACK(rOF, 0, 0) | cz_fse_code_bits(sh.b.c.llml, 1, (uint32_t)rOF)) : 0;
    const uint32_t fML = rML >= 0 ? (CZ_FSE_PACK(rML, 0, 0) | cz_fse_code_bits(sh.b.c.llml, 2, (uint32_t)rML)) : 0;
    const int any_rle = (rLL >= 0) | (rOF >= 0) | (rML >= 0);
    const uint32_t* TLL = CZ_FSE_LL; const uint32_t* TOF = CZ_FSE_OF; const uint32_t* TML = CZ_FSE_ML;
    /* stage the top two 1 KiB blocks of the stream */
    uintptr_t loaded_lo;
    {
        const uintptr_t top = (((uintptr_t)E - (E > S ? 1 : 0)) & ~(uintptr_t)(CZ_RING_BLOCK - 1));
        cz_ring_load_block(S, E, top); cz_ring_load_block(S, E, top - CZ_RING_BLOCK);
        loaded_lo = top - CZ_RING_BLOCK;
    }
    __syncthreads();
    int32_t pos = (int32_t)(E - S) * 8;                                 /* bits_remaining, lane 0 is authoritative */
    uint32_t sLL = 0, sOF = 0, sML = 0;
    if (LANE == 0) {
        int e = 0, skipped = 0;
        for (;;) {                                                      /* padding :46-64 */
            const uint32_t b = pos > 0 ? cz_field(cz_ring_window(sbits, pos - 1), 0, 1) : 0; pos -= 1; skipped++;
            if (b == 1 || skipped > 8) break;
        }
        if (skipped > 8) e = CZ_E_SEQ_EXTRA_PADDING;
        /* init order LL, OF, ML (:207-218); a table that was never set is TableIsUninitialized */
        const uint32_t logs[3] = { sh.fse_log[0], sh.fse_log[1], sh.fse_log[2] };
        const int32_t rl[3] = { rLL, rOF, rML }; uint32_t stv[3] = {0, 0, 0};
        for (int t = 0; t < 3 && !e; t++) {
            if (rl[t] >= 0) continue;
            if (!logs[t]) { e = CZ_E_SEQ_TABLE_UNINIT; break; }
            stv[t] = pos > 0 ? cz_field(cz_ring_window(sbits, pos - 1), 0, logs[t]) : 0; pos -= (int32_t)logs[t];
        }
        sLL = stv[0]; sOF = stv[1]; sML = stv[2];
        bc.chunk_err = e;
    }
    __syncthreads();
    { const int e = cz_unii(bc.chunk_err); __syncthreads(); if (e) return e; }
    CZ_PROF_ACC(CZ_P_RING);
    int exec_err = 0;                                                   /* first execution error, reported only if the
                                                                           rest of the section decodes (reference order) */
    uint32_t h0 = cz_uni(sh.hist[0]), h1 = cz_uni(sh.hist[1]), h2 = cz_uni(sh.hist[2]);   /* uniform copy in every lane */
    for (uint32_t done = 0; done < nseq; done += 64) {
        const uint32_t cnt = nseq - done < 64 ? nseq - done : 64;
        /* keep CZ_RING_NEED bytes below the cursor staged */
        {
            const int32_t p0 = cz_unii(__shfl(pos, 0));
            const intptr_t cur = (intptr_t)S + ((p0 > 0 ? p0 - 1 : 0) >> 3);
            if (cur - (intptr_t)CZ_RING_NEED < (intptr_t)loaded_lo) {
                loaded_lo -= CZ_RING_BLOCK;
                cz_ring_load_block(S, E, loaded_lo);
                __syncthreads();
            }
        }
        CZ_PROF_ACC(CZ_P_RING);
        if (LANE == 0) {
            /* Fast pass: straight-line, no per-sequence branches; any invalid code or overrun only
               sets a flag, and the chunk is then redone by the careful loop below, which finds the
               first failing sequence and its reference error code. */
            const int32_t pos_save = pos; const uint32_t sLL_save = sLL, sOF_save = sOF, sML_save = sML;
            const int is_last_chunk = done + cnt >= nseq;
            const uint32_t full = is_last_chunk ? cnt - 1 : cnt;
            uint32_t bad = 0; int32_t neg = 0; uint32_t slow = 0;
                        /* u = ring-space bit address just above the next unread bit; the three dwords at and
               below u>>5 hold the next 64+ stream bits (the ring is mirrored 8 bytes below its
               start, so ba-4 / ba-8 never wrap). */
            int32_t u = (int32_t)sbits + pos;
            const uint8_t* ringb = sh.a.t4.ring;
            for (uint32_t i = 0; i < full; i++) {
                const uint32_t ba = ((uint32_t)u >> 3) & (CZ_RING_BYTES - 4);
                const uint32_t w2 = *(const uint32_t*)(ringb + ba), w1 = *(const uint32_t*)(ringb + ba - 4), w0 = *(const uint32_t*)(ringb + ba - 8);
                uint32_t eLL = TLL[sLL], eOF = TOF[sOF], eML = TML[sML];
                if (any_rle) { eLL = rLL >= 0 ? fLL : eLL; eOF = rOF >= 0 ? fOF : eOF; eML = rML >= 0 ? fML : eML; }
                sh.a.t4.rec_pos[i] = u - (int32_t)sbits; sh.a.t4.rec_st[i] = sLL | (sOF << 9) | (sML << 18);
                const uint32_t sum = eLL + eOF + eML, a = sum & 0x7F, nbs = (sum >> 7) & 0x3F;
                bad |= eLL | eOF | eML;
                slow |= a > 32;                                         /* rare: more than 32 extra bits -> careful loop */
                /* 32 stream bits that follow the `a` extra bits (read order: extras first, :239) */
                const uint32_t ph = (uint32_t)u & 31, sel = ph >= a;
                const uint32_t xh = __builtin_amdgcn_alignbit(sel ? w2 : w1, sel ? w1 : w0, (ph - a) & 31);
                const uint32_t nl = CZ_FSE_NB(eLL), nm = CZ_FSE_NB(eML), no = CZ_FSE_NB(eOF);
                sLL = CZ_FSE_BASE(eLL) + __builtin_amdgcn_ubfe(xh, 32 - nl, nl);         /* update order LL, ML, OF (:258-277) */
                sML = CZ_FSE_BASE(eML) + __builtin_amdgcn_ubfe(xh, 32 - nl - nm, nm);
                sOF = CZ_FSE_BASE(eOF) + __builtin_amdgcn_ubfe(xh, 32 - nl - nm - no, no);
                u -= (int32_t)(a + nbs);
                neg |= u - (int32_t)sbits;
            }
            pos = u - (int32_t)sbits;
            if (!slow && is_last_chunk) {                               /* the block's last sequence: no state update */
                uint32_t eLL = TLL[sLL & 1023], eOF = TOF[sOF & 1023], eML = TML[sML & 1023];
                if (any_rle) { eLL = rLL >= 0 ? fLL : eLL; eOF = rOF >= 0 ? fOF : eOF; eML = rML >= 0 ? fML : eML; }
                sh.a.t4.rec_pos[cnt - 1] = pos; sh.a.t4.rec_st[cnt - 1] = sLL | (sOF << 9) | (sML << 18);
                bad |= eLL | eOF | eML;
                pos -= (int32_t)((eLL + eOF + eML) & 0x7F);
                neg |= pos;
                if (pos > 0) slow = 1;                                  /* ExtraBits: let the careful loop report it */
            }
            if (!slow && !((bad >> 13) & 1) && neg >= 0) bc.chunk_err = 0;
            else {
            pos = pos_save; sLL = sLL_save; sOF = sOF_save; sML = sML_save;
            int e = 0;
            for (uint32_t i = 0; i < cnt; i++) {                        /* :223-286, serial core (careful) */
                const uint64_t W = pos > 0 ? cz_ring_window(sbits, pos - 1) : 0;
                const uint32_t eLL = rLL >= 0 ? fLL : TLL[sLL], eOF = rOF >= 0 ? fOF : TOF[sOF], eML = rML >= 0 ? fML : TML[sML];
                sh.a.t4.rec_pos[i] = pos; sh.a.t4.rec_st[i] = sLL | (sOF << 9) | (sML << 18);
                if (CZ_FSE_INV(eOF)) { e = CZ_E_SEQ_UNSUPPORTED_OFFSET; break; }          /* :235 */
                if (CZ_FSE_INV(eLL) | CZ_FSE_INV(eML)) { e = CZ_E_SEQ_TOO_MANY_BITS; break; } /* num_bits 255 -> TooManyBits :239 */
                const uint32_t a = CZ_FSE_XB(eOF) + CZ_FSE_XB(eML) + CZ_FSE_XB(eLL);     /* extra bits, read first (:239) */
                if (done + i + 1 < nseq) {                              /* :258-277 state updates, order LL, ML, OF */
                    const uint32_t nl = CZ_FSE_NB(eLL), nm = CZ_FSE_NB(eML), no = CZ_FSE_NB(eOF), tot = a + nl + nm + no;
                    uint64_t V = W; uint32_t o = a;
                    if (tot > 64) { V = pos - (int32_t)a > 0 ? cz_ring_window(sbits, pos - (int32_t)a - 1) : 0; o = 0; }
                    sLL = CZ_FSE_BASE(eLL) + cz_field(V, o, nl);
                    sML = CZ_FSE_BASE(eML) + cz_field(V, o + nl, nm);
                    sOF = CZ_FSE_BASE(eOF) + cz_field(V, o + nl + nm, no);
                    pos -= (int32_t)tot;
                } else pos -= (int32_t)a;
                if (pos < 0) { e = CZ_E_SEQ_NOT_ENOUGH_BYTES; break; }  /* :281 */
            }
            if (!e && done + cnt >= nseq && pos > 0) e = CZ_E_SEQ_EXTRA_BITS;           /* :292 */
            bc.chunk_err = e;
            }
        }
        __syncthreads();
        { const int e = cz_unii(bc.chunk_err); __syncthreads(); if (e) return e; }
        CZ_PROF_ACC(CZ_P_CHAIN);
        if (!exec_err) {
            /* every lane finishes its own sequence: extra bits -> (ll, ml, offset_value) */
            uint32_t ll = 0, ml = 0, ov = 4;
            const int active = (uint32_t)LANE < cnt;
            if (active) {
                const int32_t p = sh.a.t4.rec_pos[LANE]; const uint32_t st = sh.a.t4.rec_st[LANE];
                const uint32_t eLL = rLL >= 0 ? fLL : TLL[st & 511], eOF = rOF >= 0 ? fOF : TOF[(st >> 9) & 511], eML = rML >= 0 ? fML : TML[(st >> 18) & 511];
                const uint64_t W = p > 0 ? cz_ring_window(sbits, p - 1) : 0;
                const uint32_t oc = CZ_FSE_XB(eOF), mx = CZ_FSE_XB(eML), lx = CZ_FSE_XB(eLL);
                const uint32_t tl = sh.b.c.llml[CZ_FSE_SYM(eLL)], tm = sh.b.c.llml[40 + CZ_FSE_SYM(eML)];
                ov = (1u << oc) + cz_field(W, 0, oc);                   /* :243 */
                ml = (tm & 0xFFFFFFu) + cz_field(W, oc, mx);            /* :249-256 */
                ll = (tl & 0xFFFFFFu) + cz_field(W, oc + mx, lx);
            }
            exec_err = cz_history_and_execute(x, lit, cnt, ll, ml, ov, h0, h1, h2);
            CZ_PROF_T0();
        }
        __syncthreads();
    }
    if (exec_err) return exec_err;
    if (LANE == 0) { sh.hist[0] = h0; sh.hist[1] = h1; sh.hist[2] = h2; }
    /* remaining literals (sequence_execution.cairo:72-78) */
    if (x.lit_used < lit.len) {
        const uint32_t rest = lit.len - x.lit_used;
        if (x.produced + rest > x.cap) return CZ_E_OUTPUT_TOO_SMALL;
        cz_lit_coop_copy(x.out + x.produced, lit, x.lit_used, rest);
        x.produced += rest;
    }
    __syncthreads();
    CZ_PROF_ACC(CZ_P_LITCOPY);
    return 0;
}

#endif /* !CZ_EXEC_ONLY */
/* Same as cz_sequences, for a block whose FSE chain was already run by cz_chain_kernel: every
 * sequence has an 8-byte record in the chain arena — low word = the 32 stream bits that start with
 * the sequence's extra bits (OF, ML, LL; sequence_section_decoder.cairo:239-256), high word = LL state |
 * ML state << 9 | OF state << 18 — and the block header carries the state->code byte maps of its LL,
 * ML and OF tables.  This pass needs neither decoding tables nor the bitstream, only the maps (kept in
 * the LDS that holds the FSE tables otherwise; they persist over Repeat-mode blocks). */
#define CZ_CHAIN_MAP_WORDS 160u   /* LL 512 B, ML 512 B, OF 256 B */
#define CZC_REC_WIDE 0x80000000u /* record of a sequence with more than 32 extra bits: low word = bits of the stream still unread before it */
/* 64 stream bits below bit p of the reversed bitstream that starts at S (zero below bit 0); rare path of the record-driven decode */
__device__ static inline uint64_t cz_stream_window64(cz_gcptr S, uint32_t p) {
    if (p == 0) return 0;
    const int32_t hb = (int32_t)((p - 1) >> 3);
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) if (hb - k >= 0) v |= (uint64_t)S[hb - k] << (56 - 8 * k);
    const uint32_t drop = 7u - ((p - 1) & 7u);
    if (drop) { const uint32_t lo = hb - 8 >= 0 ? S[hb - 8] : 0u; v = (v << drop) | (lo >> (8 - drop)); }
    return v;
}
/* ---- the record-driven chunk loops -----------------------------------------------------------------------------------
 * cz_sequences_rec_general  the loop over chunks of 64 records in its general form: cz_chunk_plan + cz_chunk_copy per chunk (all
 *                           checks in the reference's order, dictionary reach, long runs, wide records, partial chunks).
 * cz_sequences_rec_fast     the same loop for frames whose positions fit 32 bits, as long as the chunks are full chunks of short
 *                           sequences: everything uniform lives in scalar registers, positions are 32-bit offsets from two scalar
 *                           bases, the two prefix sums are one packed scan, and the chunk is assembled in LDS without a predicate
 *                           per match byte (cz_fast_group).  It stops at a chunk it cannot take: that chunk and the next few go
 *                           through the general loop (cz_sequences_rec), then it is entered again.
 * Each is a function of its own (not inlined), and neither calls the other: the register allocation of one does not pay for the
 * other's live values. */
/* the state -> code maps of the tables the block defined, from its arena header into the LDS that holds FSE tables otherwise */
__device__ static inline void cz_rec_load_maps(cz_gcptr64 maps, uint32_t mapflags) {
    uint8_t* mapll = (uint8_t*)CZ_FSE_LL; uint8_t* mapof = mapll + 1024;
    __syncthreads();
    {
        const uint32_t half = (uint32_t)LANE >> 5, j = (uint32_t)LANE & 31;           /* 32 lanes x 16 B per 512-byte map */
        if ((mapflags >> (half ? 2 : 0)) & 1u) { uint4 v; __builtin_memcpy(&v, (cz_gcptr)maps + 512u * half + 16u * j, 16); *(uint4*)(mapll + 512u * half + 16u * j) = v; }
        if (((mapflags >> 1) & 1u) && LANE < 16) { uint4 v; __builtin_memcpy(&v, (cz_gcptr)maps + 1024u + 16u * (uint32_t)LANE, 16); *(uint4*)(mapof + 16u * (uint32_t)LANE) = v; }
    }
    __syncthreads();
}
__device__ static inline uint32_t cz_rec_values(uint64_t r, cz_gcptr bits, uint32_t& ll, uint32_t& ml) {
    const uint8_t* mapll = (const uint8_t*)CZ_FSE_LL; const uint8_t* mapml = mapll + 512; const uint8_t* mapof = mapll + 1024;
    const uint32_t xt = (uint32_t)r, st = (uint32_t)(r >> 32);
    const uint32_t oc = mapof[(st >> 18) & 255];
    const uint32_t tl = sh.b.c.llml[mapll[st & 511]], tm = sh.b.c.llml[40 + mapml[(st >> 9) & 511]];
    const uint32_t mx = tm >> 24, lx = tl >> 24;
    uint32_t ov;
    if (!(st & CZC_REC_WIDE)) {                                         /* <= 32 extra bits: they are the top of the record's low word */
        ov = (1u << oc) + __builtin_amdgcn_ubfe(xt, 32 - oc, oc);       /* :243 */
        ml = (tm & 0xFFFFFFu) + __builtin_amdgcn_ubfe(xt, 32 - oc - mx, mx);     /* :249-256 */
        ll = (tl & 0xFFFFFFu) + __builtin_amdgcn_ubfe(xt, 32 - oc - mx - lx, lx);
    } else {                                                            /* the low word says where the extra bits are in the bitstream */
        const uint64_t W = cz_stream_window64(bits, xt);
        ov = (1u << oc) + cz_field(W, 0, oc);
        ml = (tm & 0xFFFFFFu) + cz_field(W, oc, mx);
        ll = (tl & 0xFFFFFFu) + cz_field(W, oc + mx, lx);
    }
    return ov;
}
/* chunks [first, end) of a block (end: a multiple of 64 past first, or nseq); history in sh.hist, positions in xref.  The execution
   context is worked on in registers (wave-uniform) and written back once. */
__device__ static __attribute__((noinline)) int cz_sequences_rec_general(CzExecCtx& xref, const CzLit lit, cz_gcptr64 rec, uint32_t nseq, cz_gcptr bits, uint32_t first, uint32_t end) {
    CzExecCtx x = xref;
    x.out = (cz_gptr)cz_uni64((uint64_t)x.out); x.cap = cz_uni64(x.cap); x.produced = cz_uni64(x.produced); x.drained = cz_uni64(x.drained);
    x.window = cz_uni64(x.window); x.lit_used = cz_uni(x.lit_used);
    CZ_PROF_DECL; CZ_PROF_T0();
    int exec_err = 0;
    uint32_t h0 = cz_uni(sh.hist[0]), h1 = cz_uni(sh.hist[1]), h2 = cz_uni(sh.hist[2]);
    auto load_rec = [&](uint32_t at) -> uint64_t { return at + (uint32_t)LANE < nseq ? rec[at + (uint32_t)LANE] : 0; };   /* coalesced 8-byte loads */
    auto plan = [&](uint64_t r, uint32_t at, uint64_t produced, uint32_t lit_used) -> CzPlan {
        const uint32_t cnt = nseq - at < 64 ? nseq - at : 64;
        uint32_t ll = 0, ml = 0, ov = 4;
        if ((uint32_t)LANE < cnt) ov = cz_rec_values(r, bits, ll, ml);
#ifdef CZ_EXP_NOHIST   /* diagnostic only (wrong output): instruction count of the chunk loop without the repeat-offset scan */
        const uint32_t actual = ov > 3 ? ov - 3 : h0 + ov;
#else
        const uint32_t actual = cz_history(cnt, ll, ov, h0, h1, h2);
#endif
        return cz_chunk_plan(x, produced, lit_used, lit, cnt, ll, ml, actual);
    };
    /* Records are loaded three chunks ahead; each chunk is planned (codes -> values, repeat offsets,
     * positions, every check) and then copied.  (Planning a chunk ahead of the copy and touching its
     * source lines early was measured: once no access was a flat_* one it no longer paid.) */
    uint64_t r1 = load_rec(first), r2 = load_rec(first + 64), r3 = load_rec(first + 128);
    for (uint32_t done = first; done < end; done += 64) {
        const CzPlan cur = plan(r1, done, x.produced, x.lit_used);
        r1 = r2; r2 = r3; r3 = load_rec(done + 192);
        if (cur.err > 0) return cur.err;
        CZ_PROF_ACC(CZ_P_EXTRACT);
#ifdef CZ_EXP_NOCOPY   /* diagnostic only (wrong output): ... without the copy stage */
        x.produced += cur.sum_tot; x.lit_used += cur.sum_ll; exec_err = 0;
#else
        exec_err = cz_chunk_copy(x, lit, cur);
#endif
        CZ_PROF_T0();
        if (exec_err) return exec_err;
        cz_wave_sync();
    }
    if (LANE == 0) { sh.hist[0] = h0; sh.hist[1] = h1; sh.hist[2] = h2; }
    xref.produced = x.produced; xref.lit_used = x.lit_used;
    return 0;
}
/* unaligned loads from global memory (the sizes are literals: a size that depends on a template parameter makes the builtin an ordinary call) */
__device__ static inline uint32_t cz_ldu16(cz_gcptr p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ static inline uint32_t cz_ldu32(cz_gcptr p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ static inline uint64_t cz_ldu64(cz_gcptr p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ static inline uint4 cz_ldu128(cz_gcptr p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
/* First bytes of every lane's literal run and match into the chunk buffer: NL literal bytes, NMB match bytes, every load
 * before the first LDS write.  Match bytes go first, from the LAST byte down, without a predicate: a byte a lane writes beyond
 * its own match lands on a position whose true owner — a later sequence's match byte of lower index, or a literal byte —
 * is written after it (LDS writes of one wave execute in program order), so the true bytes win.  Lanes whose match reads this
 * chunk's own output write what they loaded all the same; the dependency rounds redo them.  Literal bytes follow, each
 * with its own predicate (address select to a dump byte).  Source addresses are 32-bit offsets from two uniform bases. */
template <int NL, int NMB>
__device__ static inline void cz_fast_group(uint8_t* ob, uint32_t orel, uint32_t drel, uint32_t ll, cz_gcptr lbase, uint32_t loff, int lit_rle, uint32_t rle_word,
                                            cz_gcptr obase, uint32_t moff) {
    uint32_t lw[2] = {rle_word, rle_word}, mw[4] = {0, 0, 0, 0};
    if (!lit_rle) {
        if (NL <= 2) lw[0] = cz_ldu16(lbase + loff);
        else if (NL <= 4) lw[0] = cz_ldu32(lbase + loff);
        else { const uint64_t t = cz_ldu64(lbase + loff); lw[0] = (uint32_t)t; lw[1] = (uint32_t)(t >> 32); }
    }
    if (NMB == 4) mw[0] = cz_ldu32(obase + moff);
    else if (NMB == 8) { const uint64_t t = cz_ldu64(obase + moff); mw[0] = (uint32_t)t; mw[1] = (uint32_t)(t >> 32); }
    else { const uint4 t = cz_ldu128(obase + moff); mw[0] = t.x; mw[1] = t.y; mw[2] = t.z; mw[3] = t.w; }
    uint8_t* const md = ob + drel;
#pragma unroll
    for (int j = NMB - 1; j >= 0; j--) { md[j] = (uint8_t)(mw[j >> 2] >> (8 * (j & 3))); CZ_LOCKSTEP(); }
    const uint32_t dump = CZ_OBUF_BYTES + 16 + (uint32_t)LANE;
#pragma unroll
    for (int j = NL - 1; j >= 0; j--) { ob[((uint32_t)j < ll ? orel : dump) + (uint32_t)j] = (uint8_t)(lw[j >> 2] >> (8 * (j & 3))); CZ_LOCKSTEP(); }
}
/* Chunks [first_, nseq) of a block.  Returns 0 (block done: the literals after the last sequence are copied too), a status, or -2:
   the chunk at sh.rec_next needs the general form — positions in xref, history in sh.hist.  The loop makes no call (a call
   inside it would keep its live values in the callee-saved half of the registers: 113 spills at 96 registers, none like this). */
__device__ static __attribute__((noinline)) int cz_sequences_rec_fast(CzExecCtx& xref, const CzLit lit, cz_gcptr64 maps, cz_gcptr64 rec_, uint32_t nseq_, uint32_t mapflags, cz_gcptr bits_, uint32_t first_) {
    CZ_PROF_DECL; CZ_PROF_T0();
    cz_rec_load_maps(maps, mapflags);
    /* uniform state of the block, in scalar registers */
    const uint32_t nseq = cz_uni(nseq_);
    cz_gcptr64 rec = (cz_gcptr64)cz_uni64((uint64_t)(uintptr_t)rec_);
    cz_gcptr bits = (cz_gcptr)cz_uni64((uint64_t)(uintptr_t)bits_);
    cz_gptr obase = (cz_gptr)cz_uni64((uint64_t)(uintptr_t)xref.out);
    cz_gcptr lbase = (cz_gcptr)cz_uni64((uint64_t)(uintptr_t)lit.p);
    const uint32_t lit_len = cz_uni(lit.len), rle_word = 0x01010101u * cz_uni((uint32_t)lit.byte);
    const int lit_rle = cz_unii((int)lit.rle);
    const uint32_t cap = cz_uni((uint32_t)xref.cap);
    uint32_t P = cz_uni((uint32_t)xref.produced), lit_used = cz_uni(xref.lit_used);
    uint32_t h0 = cz_uni(sh.hist[0]), h1 = cz_uni(sh.hist[1]), h2 = cz_uni(sh.hist[2]);
    uint8_t* const ob = sh.a.t4.obuf;
#if defined(CZ_EXP_NT_REC) && defined(__HIP_DEVICE_COMPILE__)
    auto load_rec = [&](uint32_t first) -> uint64_t { const uint32_t i = first + (uint32_t)LANE; return __builtin_nontemporal_load(&rec[i < nseq ? i : nseq - 1]); };   /* (diagnostic: records are read once — keep them out of the caches' way) */
#else
    auto load_rec = [&](uint32_t first) -> uint64_t { const uint32_t i = first + (uint32_t)LANE; return rec[i < nseq ? i : nseq - 1]; };   /* coalesced 8-byte loads */
#endif
    uint32_t done = cz_uni(first_);
    uint64_t r1 = load_rec(done), r2 = load_rec(done + 64), r3 = load_rec(done + 128);
    for (; done < nseq; done += 64) {
        const uint64_t r = r1;
        r1 = r2; r2 = r3; r3 = load_rec(done + 192);
        int took = 0;
        if (nseq - done >= 64 && !__ballot((uint32_t)(r >> 32) & CZC_REC_WIDE)) {
            const uint32_t s0 = h0, s1 = h1, s2 = h2;                   /* the history is put back when the chunk is left to the general form */
            uint32_t ll, ml;
            const uint32_t ov = cz_rec_values(r, bits, ll, ml);
            const uint32_t off = cz_history(64u, ll, ov, h0, h1, h2);
            CZ_PROF_ACC(CZ_P_EXTRACT);
            /* both prefix sums in one scan: literal lengths in the low half, output lengths in the high half (runs of at most
               8 / 16 bytes here: the sums stay far below 65 536) */
            const uint32_t tot = ll + ml;
            const int small = !__ballot(ll > 8u || ml > 16u);
            const uint32_t pk = ll | (tot << 16), incl = cz_wave_incl_scan(pk), sums = cz_readlane(incl, 63);
            const uint32_t sum_ll = sums & 0xFFFFu, sum_tot = sums >> 16;
            const uint32_t excl = incl - pk, lrel = excl & 0xFFFFu, orel = excl >> 16, drel = orel + ll;
            const uint32_t d = P + drel;                                /* where the match goes */
            /* sequence_execution.cairo:28-36 (literals), :47 (zero offset), decode_buffer.cairo:65 (offset beyond the output so far),
               and the capacity of the caller's buffer; the 8-byte literal loads must stay inside the literal buffer */
            const int bad = (off - 1u >= d) | (d + ml > cap);
            if (small && sum_tot <= CZ_OBUF_BYTES && !__ballot(bad) && lit_used + sum_ll + 8u <= lit_len) {
                const uint32_t span = off < ml ? off : ml;
                const int32_t srel = (int32_t)drel - (int32_t)off, send = srel + (int32_t)span;   /* source range relative to the chunk, clipped to the match's own start */
                const int near = send > 0;
                if (!__ballot(!near && off < ml)) {                     /* (a far match that overlaps itself: the general form does the period) */
                    const uint32_t loff = lit_used + lrel, moff = d - off;
                    const unsigned long long big = __ballot(ll > 4u || ml > 8u), mid = __ballot(ll > 2u || ml > 4u);
                    if (!mid) cz_fast_group<2, 4>(ob, orel, drel, ll, lbase, loff, lit_rle, rle_word, (cz_gcptr)obase, moff);
                    else if (!big) cz_fast_group<4, 8>(ob, orel, drel, ll, lbase, loff, lit_rle, rle_word, (cz_gcptr)obase, moff);
                    else cz_fast_group<8, 16>(ob, orel, drel, ll, lbase, loff, lit_rle, rle_word, (cz_gcptr)obase, moff);
                    cz_wave_sync();
                    CZ_PROF_ACC(CZ_P_LITCOPY);
                    /* matches that read this chunk's output: rounds.  W = destination of the first undone match; a match may go
                       once its source range (clipped to its own destination) lies below W. */
                    cz_gcptr cout = (cz_gcptr)obase + P;
                    int undone = near;
                    for (;;) {
                        const unsigned long long pend = __ballot(undone);
                        if (!pend) break;
                        const int32_t W = (int32_t)cz_readlane(drel, cz_unii(__ffsll((long long)pend) - 1));
                        if (undone && send <= W) {
                            uint32_t idx = 0;
                            for (uint32_t k = 0; k < ml; k++) {
                                const int32_t q = srel + (int32_t)idx;
                                ob[drel + k] = q < 0 ? cout[q] : ob[q];
                                idx = idx + 1 == off ? 0 : idx + 1;
                            }
                            undone = 0;
                        }
                        cz_wave_sync();
                    }
                    /* write the assembled chunk: 4 bytes per lane per pass (the global address need not be aligned).  Later loads
                       of these bytes by this wave are ordered behind the stores by the memory pipeline. */
                    for (uint32_t i = 4u * (uint32_t)LANE; i < sum_tot; i += 256) {
                        if (i + 4 <= sum_tot) { const uint32_t w = *(const uint32_t*)(ob + i); __builtin_memcpy(obase + (P + i), &w, 4); }
                        else for (uint32_t j = i; j < sum_tot; j++) obase[P + j] = ob[j];
                    }
                    cz_wave_sync();
                    CZ_PROF_ACC(CZ_P_MATCH);
                    P += sum_tot; lit_used += sum_ll; took = 1;
                    CZ_PROF_CNT(CZ_P_N_FAST);
                }
            }
            if (!took) { h0 = s0; h1 = s1; h2 = s2; }
        }
        if (!took) {
            cz_wave_sync();
            if (LANE == 0) { sh.hist[0] = h0; sh.hist[1] = h1; sh.hist[2] = h2; sh.rec_next = done; }
            cz_wave_sync();
            xref.produced = P; xref.lit_used = lit_used;
            return -2;
        }
    }
    cz_wave_sync();
    if (LANE == 0) { sh.hist[0] = h0; sh.hist[1] = h1; sh.hist[2] = h2; }
    xref.produced = P; xref.lit_used = lit_used;
    if (lit_used < lit_len) {                                           /* sequence_execution.cairo:72-78 */
        const uint32_t rest = lit_len - lit_used;
        if (xref.produced + rest > xref.cap) return CZ_E_OUTPUT_TOO_SMALL;
        cz_lit_coop_copy(xref.out + xref.produced, lit, lit_used, rest);
        xref.produced += rest; xref.lit_used = lit_len;
    }
    __syncthreads();
    return 0;
}

/* One block whose sequences cz_chain_kernel left as records (header: maps of the tables it defined, then the records).  A chunk
   the fast loop cannot take goes through the general loop on its own; after CZ_FAST_MISSES of those the general loop takes the
   rest of the frame.  (Handing blocks back to the fast loop — after every chunk, after stretches of 2..8 chunks, or after 4..16
   chunks in a row that it would have taken — was measured on the corpus-like mix: every variant lost, 8.0-12.8 ms against 7.6.) */
#ifndef CZ_FAST_MISSES
#define CZ_FAST_MISSES 6u
#endif
__device__ static int cz_sequences_rec(CzExecCtx& x, const CzLit lit, cz_gcptr64 maps, cz_gcptr64 rec, uint32_t nseq, uint32_t mapflags, cz_gcptr bits) {
    CZ_PROF_DECL; CZ_PROF_T0();
    /* the fast loop wants every position of the frame in 32 bits, no drained bytes and no dictionary content (then offset <=
       position is the whole reach test of decode_buffer.cairo:62-93) */
    uint32_t first = 0;
#ifndef CZ_EXP_NOFAST
    if (cz_uni64(x.cap) < 0xC0000000ull && cz_uni64(x.drained) == 0 && (sh.dict_len[0] | sh.dict_len[1]) == 0 && cz_uni(sh.rec_misses) <= CZ_FAST_MISSES) {
        uint32_t flags = mapflags;
        for (;;) {
            const int e = cz_sequences_rec_fast(x, lit, maps, rec, nseq, flags, bits, first);   /* (it loads the maps, and copies the literals after the last sequence) */
            if (e != -2) return e;
            first = cz_uni(sh.rec_next); flags = 0;
            cz_wave_sync();
            if (LANE == 0) sh.rec_misses += 1;
            cz_wave_sync();
            if (cz_uni(sh.rec_misses) > CZ_FAST_MISSES) break;
            const uint32_t end = nseq - first > 64u ? first + 64u : nseq;
            const int e2 = cz_sequences_rec_general(x, lit, rec, nseq, bits, first, end);
            if (e2) return e2;
            cz_wave_sync();
            first = end;
        }
    } else
#endif
    cz_rec_load_maps(maps, mapflags);
    CZ_PROF_ACC(CZ_P_RING);
    const int e = cz_sequences_rec_general(x, lit, rec, nseq, bits, first, nseq);
    if (e) return e;
    if (x.lit_used < lit.len) {                                         /* sequence_execution.cairo:72-78 */
        const uint32_t rest = lit.len - x.lit_used;
        if (x.produced + rest > x.cap) return CZ_E_OUTPUT_TOO_SMALL;
        cz_lit_coop_copy(x.out + x.produced, lit, x.lit_used, rest);
        x.produced += rest;
    }
    __syncthreads();
    CZ_PROF_ACC(CZ_P_LITCOPY);
    return 0;
}

/* ------------------------------------------------------------------ one compressed block */
/* decompress_block (block_decoder.cairo:139-235).  All lanes; uniform status. */
/* Literal nodes of a frame.  Decoding: arena + cursor = node of the next Huffman-coded block (cursor 0 = this frame has
 * none: decode literals here). */
struct CzLitPass { cz_gptr arena; uint64_t cursor; int on; uint32_t pre_blocks; };   /* on: cz_huf_kernel decoded this frame's Huffman literals; cursor: node of the
                                                                                        next such block that has one; pre_blocks: leading blocks whose output is already there */
__device__ static int cz_decompress_block(cz_gcptr blk, uint32_t bsize, CzExecCtx& x, cz_gptr lit_scratch,
                                           cz_gptr16 huf_global, int last_block, cz_gcptr64 arena, uint64_t& chain_cursor, CzLitPass& lp, int predone) {
    CzBroadcast& bc = sh.bc;
    CZ_PROF_DECL; CZ_PROF_T0();
    /* stage the head of the block for the serial header / tree parsers */
    const uint32_t stage_hi = bsize < 512 ? bsize : 512;
    for (uint32_t i = (uint32_t)LANE; i < stage_hi; i += 64) sh.a.t1.stage[i] = blk[i];
    __syncthreads();
    const int have_literals = lp.on;
    if (LANE == 0) bc.err = cz_parse_sections(blk, bsize, stage_hi, have_literals);
    __syncthreads();
#ifndef CZ_EXEC_ONLY
    if (cz_unii(bc.err) == CZ_PARSE_NEED_WTAB) {                        /* an FSE-compressed tree description: its table by the wave, then the rest of the parse */
        cz_huf_weight_table();
        __syncthreads();
        if (LANE == 0) bc.err = cz_parse_sections(blk, bsize, stage_hi, have_literals, 1);
        __syncthreads();
    }
#endif
    { const int e = cz_unii(bc.err); __syncthreads(); if (e) return e; }   /* read, then fence the slot before it is rewritten */
    CZ_PROF_ACC(CZ_P_OTHER);                                            /* (diagnostic) the serial section parse, apart from the table fill */
#ifndef CZ_EXEC_ONLY
    if (bc.huf_fill) {
        cz_huf_rank_wave(cz_uni(bc.huf_nsym)); __syncthreads();
        cz_huf_fill(bc.huf_nsym); __syncthreads();
        if (cz_uni(bc.regen) >= 2048u || !last_block) { cz_huf_fill_multi(); __syncthreads(); }   /* pays from a few symbols per lane on; a carried table always has it */
        if (!last_block) {                                              /* carried for later Treeless blocks */
            for (uint32_t i = (uint32_t)LANE; i < 1024; i += 64) ((CZ_GLOBAL uint32_t*)huf_global)[i] = ((const uint32_t*)sh.a.huf)[i];
        }
    } else if (bc.lit_type == 3 && !have_literals) {                    /* Treeless: bring the carried table back */
        for (uint32_t i = (uint32_t)LANE; i < 1024; i += 64) ((uint32_t*)sh.a.huf)[i] = ((CZ_GLOBAL const uint32_t*)huf_global)[i];
        __syncthreads();
    }
#endif
    CZ_PROF_ACC(CZ_P_HUFBUILD);
    /* literals */
    const uint32_t regen = cz_uni(bc.regen), lit_total = cz_uni(bc.lit_total), nseq = cz_uni(bc.nseq);
    const int seq_hdr_err = cz_unii(bc.seq_hdr_err);
    CzLit lit; lit.rle = 0; lit.byte = 0; lit.len = regen; lit.p = blk;
    const uint32_t lt = cz_uni(bc.lit_type), nseq_early = seq_hdr_err ? 1u : nseq;
    if (predone) {
        /* a block without sequences ahead of the frame's first block with sequences: cz_tile_kernel / cz_huf_kernel already put its
           literals — its whole output (block_decoder.cairo:229-232) — where they belong; the scan checked the capacity */
        if (seq_hdr_err || nseq) return CZ_E_INVALID_ARG;               /* cannot happen: the scan read the same headers */
        x.produced += regen;
        __syncthreads();
        return 0;
    }
    if (lt == 0) { lit.p = blk + (lit_total - regen); }           /* Raw: used in place (literals_section_decoder.cairo:39-42) */
    else if (lt == 1) { lit.rle = 1; lit.byte = blk[lit_total - 1]; } /* RLE :43-46 */
    else if (have_literals) {
        /* decoded by cz_huf_kernel: node = {next, regen, 0, bytes} */
        if (lp.cursor < 64) return CZ_E_INVALID_ARG;                    /* cannot happen: the scan laid out a node for this block */
        CZ_GLOBAL const uint64_t* node = (CZ_GLOBAL const uint64_t*)(lp.arena + lp.cursor);
        const uint64_t nxt = node[0], meta = node[1];
        if ((uint32_t)meta != regen) return CZ_E_INVALID_ARG;           /* cannot happen: both passes read the same header */
        lit.p = (cz_gcptr)(lp.arena + lp.cursor + 16);
        lp.cursor = cz_uni64(nxt);
        if (nseq_early == 0) {
            if (x.produced + regen > x.cap) return CZ_E_OUTPUT_TOO_SMALL;
            cz_coop_copy(x.out + x.produced, lit.p, regen);
        }
    } else {
#ifdef CZ_EXEC_ONLY
        return CZX_FALLBACK;                                            /* (cz_parse_sections already said so) */
#else
        cz_gptr target = lit_scratch;
        if (0) {
        } else if (nseq_early == 0) {                                   /* no sequences: decode straight into the output */
            if (x.produced + regen > x.cap) return CZ_E_OUTPUT_TOO_SMALL;
            target = x.out + x.produced;
        } else if (regen > CZ_LIT_SCRATCH_BYTES) return CZ_E_UNSUPPORTED;
        const int e = cz_decode_huf_literals(blk, target);
        if (e) return e;
        lit.p = target;
        __syncthreads();
#endif
    }
    CZ_PROF_ACC(CZ_P_HUFDEC);
    if (seq_hdr_err) return seq_hdr_err;                                /* block_decoder.cairo:198-204 */
    if (nseq == 0) {                                                 /* :229-232 */
        if (lt < 2) {
            if (x.produced + lit.len > x.cap) return CZ_E_OUTPUT_TOO_SMALL;
            cz_lit_coop_copy(x.out + x.produced, lit, 0, lit.len);
        }
        x.produced += lit.len;
        __syncthreads();
        return 0;
    }
    if (chain_cursor) {
        /* cz_chain_kernel already ran this block's FSE chain: header = {nseq|map flags, bitstream_off, next}, code maps, records */
        const uint64_t w0 = arena[chain_cursor], w1 = arena[chain_cursor + 1], w2 = arena[chain_cursor + 2];
        cz_gcptr64 maps = arena + chain_cursor + 4;
        cz_gcptr64 rec = maps + CZ_CHAIN_MAP_WORDS;
        chain_cursor = cz_uni64(w2);
        const uint32_t rn = cz_uni((uint32_t)(w0 >> 32)), mapflags = cz_uni((uint32_t)w0);
        if (rn != nseq) return CZ_E_INVALID_ARG;                        /* cannot happen: both passes walk the same bytes */
        x.lit_used = 0;
        CZ_PROF_ACC(CZ_P_SEQTAB);
        return cz_sequences_rec(x, lit, maps, rec, nseq, mapflags, blk + cz_uni((uint32_t)w1));
    }
#ifdef CZ_EXEC_ONLY
    return CZX_FALLBACK;                                                /* a sequences section without chain records */
#else
    /* sequence tables */
    const uint32_t so = cz_uni(bc.seq_body_off);
    /* T3: stage the table descriptions (region `a` no longer holds the T1 stage) */
    __syncthreads();
    const uint32_t st_lo = so < bsize ? so : bsize, st_hi = bsize - st_lo < 512 ? bsize : st_lo + 512;
    for (uint32_t i = st_lo + (uint32_t)LANE; i < st_hi; i += 64) sh.a.t3.stage[i - st_lo] = blk[i];
    __syncthreads();
    if (LANE == 0) bc.err = cz_parse_seq_tables(blk, bsize, st_lo, st_hi);
    __syncthreads();
    { const int e = cz_unii(bc.err); __syncthreads(); if (e) return e; }
    if (LANE < 3 && ((bc.build_mask >> LANE) & 1u)) {
        cz_fse_build(cz_fse_table(LANE), sh.a.t3.probs[LANE], bc.nprobs[LANE], bc.acc_log[LANE], sh.a.t3.counters[LANE], sh.b.c.llml, (uint32_t)LANE);
        sh.fse_log[LANE] = (uint8_t)bc.acc_log[LANE];
    }
    __syncthreads();
    x.lit_used = 0;
    CZ_PROF_ACC(CZ_P_SEQTAB);
    return cz_sequences(blk, bsize, x, lit);
#endif
}

/* ------------------------------------------------------------------ XXH64 content checksum */
/* src/utils/xxhash64.cairo:20-163 (seed 0).  The reference hashes the decoded frame as it is
 * drained (decode_buffer.cairo:162,181) and compares the low 32 bits with the 4 bytes after the
 * last block (frame_decoder.cairo:133-138).  XXH64 has four independent accumulators, each eating
 * 8 bytes of every 32-byte stripe: lanes 0..3 run them; all 64 lanes stage 512 bytes (16 stripes)
 * at a time into LDS with coalesced 8-byte loads, one block ahead. */
#define CZ_XP1 0x9E3779B185EBCA87ull
#define CZ_XP2 0xC2B2AE3D27D4EB4Full
#define CZ_XP3 0x165667B19E3779F9ull
#define CZ_XP4 0x85EBCA77C2B2AE63ull
#define CZ_XP5 0x27D4EB2F165667C5ull
__device__ static inline uint64_t cz_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ static inline uint64_t cz_xxh_round(uint64_t acc, uint64_t in) { acc += in * CZ_XP2; acc = cz_rotl64(acc, 31); return acc * CZ_XP1; }
__device__ static inline uint64_t cz_xxh_merge(uint64_t h, uint64_t v) { v = cz_xxh_round(0, v); h ^= v; return h * CZ_XP1 + CZ_XP4; }
__device__ static inline uint64_t cz_ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
/* all lanes; returns the digest in every lane */
__device__ static uint64_t cz_xxh64_frame(const uint8_t* p, uint64_t len) {
    uint64_t* stage = (uint64_t*)sh.a.huf;                              /* 2 x 512 B of the (now idle) phase region */
    uint64_t acc = LANE == 0 ? CZ_XP1 + CZ_XP2 : (LANE == 1 ? CZ_XP2 : (LANE == 2 ? 0ull : 0ull - CZ_XP1));
    const uint64_t nblk = len >> 9;                                     /* full 512-byte blocks */
    __syncthreads();
    uint64_t nxt = nblk ? cz_ld64(p + 8 * (uint64_t)LANE) : 0;
    for (uint64_t b = 0; b < nblk; b++) {
        stage[(b & 1) * 64 + (uint32_t)LANE] = nxt;
        if (b + 1 < nblk) nxt = cz_ld64(p + ((b + 1) << 9) + 8 * (uint64_t)LANE);     /* in flight during the rounds below */
        __syncthreads();
        if (LANE < 4) { const uint64_t* q = stage + (b & 1) * 64 + LANE; for (int k = 0; k < 16; k++) acc = cz_xxh_round(acc, q[4 * k]); }
    }
    __syncthreads();
    /* remaining whole stripes (< 16) and the tail, read directly */
    uint64_t off = nblk << 9;
    if (LANE < 4) for (uint64_t o = off; o + 32 <= len; o += 32) acc = cz_xxh_round(acc, cz_ld64(p + o + 8 * (uint64_t)LANE));
    off += ((len - off) >> 5) << 5;
    const uint64_t v1 = (uint64_t)__shfl((uint32_t)acc, 0) | ((uint64_t)__shfl((uint32_t)(acc >> 32), 0) << 32);
    const uint64_t v2 = (uint64_t)__shfl((uint32_t)acc, 1) | ((uint64_t)__shfl((uint32_t)(acc >> 32), 1) << 32);
    const uint64_t v3 = (uint64_t)__shfl((uint32_t)acc, 2) | ((uint64_t)__shfl((uint32_t)(acc >> 32), 2) << 32);
    const uint64_t v4 = (uint64_t)__shfl((uint32_t)acc, 3) | ((uint64_t)__shfl((uint32_t)(acc >> 32), 3) << 32);
    uint64_t h;
    if (len >= 32) {
        h = cz_rotl64(v1, 1) + cz_rotl64(v2, 7) + cz_rotl64(v3, 12) + cz_rotl64(v4, 18);
        h = cz_xxh_merge(h, v1); h = cz_xxh_merge(h, v2); h = cz_xxh_merge(h, v3); h = cz_xxh_merge(h, v4);
    } else h = CZ_XP5;
    h += len;
    const uint8_t* q = p + off; const uint8_t* end = p + len;          /* < 32 bytes, every lane redundantly */
    while (q + 8 <= end) { h ^= cz_xxh_round(0, cz_ld64(q)); h = cz_rotl64(h, 27) * CZ_XP1 + CZ_XP4; q += 8; }
    if (q + 4 <= end) { uint32_t w; __builtin_memcpy(&w, q, 4); h ^= (uint64_t)w * CZ_XP1; h = cz_rotl64(h, 23) * CZ_XP2 + CZ_XP3; q += 4; }
    while (q < end) { h ^= (uint64_t)(*q) * CZ_XP5; h = cz_rotl64(h, 11) * CZ_XP1; q++; }
    h ^= h >> 33; h *= CZ_XP2; h ^= h >> 29; h *= CZ_XP3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------ one frame */
struct CzFrameIO {
    cz_gcptr src; uint64_t src_len;
    cz_gptr dst; uint64_t dst_cap;
    uint64_t produced, drained, window;
    uint32_t parse_header, has_checksum, strategy, streaming; uint64_t strategy_n;
    uint32_t verify;          /* compute XXH64 of the decoded frame and compare with the frame's checksum */
    uint32_t pre_blocks;      /* leading blocks whose output cz_tile_kernel / cz_huf_kernel already produced */
    cz_gcptr dict; uint64_t dict_len;   /* dictionary content of the frame's DecodeBuffer (resumable path only) */
};

/* frame loop: decode_blocks (frame_decoder.cairo:156-222) / decode_from_to (:245-326) */
__device__ static int cz_run_frame(CzFrameIO io, cz_gptr lit_scratch, cz_gptr16 huf_global, CZ_GLOBAL cz_frame_result* res,
                                    cz_gcptr64 arena, uint64_t chain_cursor, CzLitPass lp) {
    CzBroadcast& bc = sh.bc;
    uint64_t pos = 0; int err = 0, hdr_ok = 0; uint32_t blocks = 0, flags = 0, cksum = 0;
    CZ_PROF_DECL; CZ_PROF_T0();
    if (io.parse_header) {
        if (LANE == 0) { bc.d0 = 0; bc.d1 = 0; bc.err = cz_parse_frame_header(io.src, io.src_len, bc); }
        __syncthreads();
        err = cz_unii(bc.err);
        if (!err) { pos = cz_uni(bc.hdr_len); io.window = cz_uni64(bc.window_size); io.has_checksum = cz_uni(bc.has_checksum); }
        __syncthreads();
    }
    CzExecCtx x; x.out = io.dst; x.cap = io.dst_cap; x.produced = io.produced; x.drained = io.drained; x.window = io.window; x.lit_used = 0;
    if (LANE == 0) { sh.rec_misses = 0; sh.dict_ptr[0] = (uint32_t)(uintptr_t)io.dict; sh.dict_ptr[1] = (uint32_t)((uint64_t)(uintptr_t)io.dict >> 32); sh.dict_len[0] = (uint32_t)io.dict_len; sh.dict_len[1] = (uint32_t)(io.dict_len >> 32); }
    __syncthreads();
    const uint64_t produced0 = io.produced;
    while (!err) {
        /* block header (block_decoder.cairo:237-321) */
        if (io.streaming && io.src_len - pos < 3) break;                /* frame_decoder.cairo:270 */
        if (LANE == 0) {
            int e = 0;
            if (io.src_len - pos < 3) e = CZ_E_BH_TRUNCATED;
            else {
                const uint8_t* p = io.src + pos;
                const uint32_t a = p[0], b = p[1], c = p[2], t = (a >> 1) & 3, size = (a >> 3) | (b << 5) | (c << 13);
                if (t == 3) e = CZ_E_BH_RESERVED;
                else if (size > 128u * 1024u) e = CZ_E_BH_SIZE_TOO_LARGE;
                bc.btype = t; bc.bsize = size; bc.blast = a & 1;
            }
            bc.err = e;
        }
        __syncthreads();
        err = cz_unii(bc.err);
        const uint32_t btype = cz_uni(bc.btype), bsize = cz_uni(bc.bsize), blast = cz_uni(bc.blast);
        __syncthreads();
        if (err) break;
        hdr_ok = 1;
        const uint64_t body = pos + 3, avail = io.src_len - body;
        const uint32_t content = btype == 1 ? 1u : bsize;
        if (io.streaming && avail < content) break;                     /* frame_decoder.cairo:282 */
        if (avail < content) { err = CZ_E_BLOCK_TRUNCATED; break; }
        const int predone = blocks < io.pre_blocks;                     /* the pre-pass kernels produced this block's output */
        if (btype == 0) {                                               /* Raw, block_decoder.cairo:97-103 */
            if (x.produced + bsize > x.cap) { err = CZ_E_OUTPUT_TOO_SMALL; break; }
            if (!predone) cz_coop_copy(x.out + x.produced, io.src + body, bsize);
            x.produced += bsize;
        } else if (btype == 1) {                                        /* RLE :104-123 */
            if (x.produced + bsize > x.cap) { err = CZ_E_OUTPUT_TOO_SMALL; break; }
            if (!predone) cz_coop_fill(x.out + x.produced, io.src[body], bsize);
            x.produced += bsize;
        } else {
            CZ_PROF_ACC(CZ_P_HDR);
            err = cz_decompress_block(io.src + body, bsize, x, lit_scratch, huf_global, (int)blast, arena, chain_cursor, lp, predone);
            CZ_PROF_T0();
            if (err) break;
        }
        __syncthreads();
        if (btype != 2) CZ_PROF_ACC(CZ_P_RAWRLE);
        pos = body + content; blocks++; hdr_ok = 0;
        if (blast) {                                                    /* frame_decoder.cairo:189-200 / :300-312 */
            flags |= CZ_RESULT_FINISHED;
            if (io.has_checksum) {
                if (io.src_len - pos >= 4) {
                    const uint8_t* p = io.src + pos;
                    cksum = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
                    flags |= CZ_RESULT_HAS_CHECKSUM; pos += 4;
                } else if (!io.streaming) err = CZ_E_CHECKSUM_TRUNCATED;
            }
            break;
        }
        if (io.strategy == 1 && blocks >= io.strategy_n) break;         /* :204-208 */
        if (io.strategy == 2 && x.produced - produced0 >= io.strategy_n) break;         /* :209-213 */
    }
#ifdef CZ_EXEC_ONLY
    if (err == CZX_FALLBACK) { __syncthreads(); return err; }
#endif
    if (err && hdr_ok) pos += 3;                                         /* the reference counts the 3 header bytes before it decodes the body (frame_decoder.cairo:172) */
    uint32_t calc = 0;
    if (io.verify && !err && (flags & CZ_RESULT_HAS_CHECKSUM) && io.produced == 0) {
        /* get_calculated_checksum == get_checksum_from_data (src/tests/decoding.cairo:16-19), on the device */
        calc = (uint32_t)cz_xxh64_frame(x.out, x.produced);
        flags |= CZ_RESULT_CHECKSUM_COMPUTED | (calc == cksum ? CZ_RESULT_CHECKSUM_MATCH : 0u);
    }
    if (LANE == 0) {
        res->status = err; res->blocks_decoded = blocks; res->bytes_consumed = pos; res->bytes_produced = x.produced;
        res->checksum_from_data = cksum; res->flags = flags; res->calculated_checksum = calc; res->reserved = 0;
        res->detail[0] = io.parse_header && (err == CZ_E_FH_SKIP_FRAME || err == CZ_E_FH_BAD_MAGIC) ? bc.d0 : blocks;
        res->detail[1] = io.parse_header && err == CZ_E_FH_SKIP_FRAME ? bc.d1 : pos;
    }
    __syncthreads();
    return err;
}

__device__ static void cz_state_reset() {                   /* scratch.cairo:23-40 */
    if (LANE == 0) {
        sh.hist[0] = 1; sh.hist[1] = 4; sh.hist[2] = 8;
        sh.fse_rle[0] = sh.fse_rle[1] = sh.fse_rle[2] = -1;
        sh.fse_log[0] = sh.fse_log[1] = sh.fse_log[2] = 0; sh.huf_max_bits = 0;
        sh.dict_lag[0] = sh.dict_lag[1] = 0;
    }
}

#ifndef CZ_EXEC_ONLY
/* DictionaryTrait::decode_dict (dictionary.cairo:35-91) on the device, with the decoder's own table builders, so that the
 * tables come out in the layout the kernels use: Huffman table, then the OF, ML and LL tables (max logs 8, 9, 9), three
 * repeat offsets, and the rest is content.  One workgroup, CZ_FSE_LDS_BYTES of dynamic LDS.
 * result: [status, offset of the content, dictionary id, magic number read]. */
extern "C" __global__ void __launch_bounds__(CZ_WG_THREADS) cz_dict_setup_kernel(const uint8_t* raw_, uint64_t len, cz_device_frame_state* st_, uint64_t* result_) {
    cz_gcptr raw = (cz_gcptr)raw_; CZ_GLOBAL cz_device_frame_state* st = (CZ_GLOBAL cz_device_frame_state*)st_;
    CZ_GLOBAL uint64_t* result = (CZ_GLOBAL uint64_t*)result_;
    CzBroadcast& bc = sh.bc;
    cz_init_llml(); cz_state_reset();
    __syncthreads();
    int err = 0; uint64_t off = 8; uint32_t magic = 0, id = 0;
    if (len < 8) err = CZ_E_DICT_TRUNCATED;                             /* (panic) :45,:50 */
    else {
        magic = (uint32_t)raw[0] | ((uint32_t)raw[1] << 8) | ((uint32_t)raw[2] << 16) | ((uint32_t)raw[3] << 24);
        id = (uint32_t)raw[4] | ((uint32_t)raw[5] << 8) | ((uint32_t)raw[6] << 16) | ((uint32_t)raw[7] << 24);
        if (magic != 0xEC30A437u) err = CZ_E_DICT_BAD_MAGIC;            /* :46-48 */
    }
    if (!err) {                                                         /* :55-62 HuffmanTable::build_decoder */
        const uint32_t left = len - off > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(len - off);
        if (LANE == 0) { uint32_t used = 0, nsym = 0; bc.err = cz_huf_read_and_rank(raw, left, 0, 0, 8, &used, &nsym, 0); bc.huf_nsym = nsym; bc.d0 = used; }
        __syncthreads();
        if (cz_unii(bc.err) == CZ_PARSE_NEED_WTAB) {
            cz_huf_weight_table();
            __syncthreads();
            if (LANE == 0) { uint32_t used = 0, nsym = 0; bc.err = cz_huf_read_and_rank(raw, left, 0, 0, 8, &used, &nsym, 1); bc.huf_nsym = nsym; bc.d0 = used; }
            __syncthreads();
        }
        err = cz_unii(bc.err);
        const uint32_t used = cz_uni((uint32_t)bc.d0), nsym = cz_uni(bc.huf_nsym);
        __syncthreads();
        if (!err && used > left) err = CZ_E_DICT_TRUNCATED;             /* (panic) slice :62 */
        if (!err) {
            cz_huf_rank_wave(nsym); __syncthreads();
            cz_huf_fill(nsym); __syncthreads();
            cz_huf_fill_multi(); __syncthreads();
            for (uint32_t i = (uint32_t)LANE; i < 1024; i += 64) ((CZ_GLOBAL uint32_t*)st->huf)[i] = ((const uint32_t*)sh.a.huf)[i];
            off += used;
        }
        __syncthreads();                                                /* region `a` becomes the T3 scratch of the table builds */
    }
    for (int o = 0; o < 3 && !err; o++) {                               /* :64-80: offsets, match lengths, literal lengths */
        const int t = o == 0 ? 1 : (o == 1 ? 2 : 0);
        const uint32_t left = len - off > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(len - off);
        if (LANE == 0) {
            CzFBits br; br.g = raw + off; br.len = left; br.idx = 0; br.stage = sh.a.t3.stage; br.stage_lo = 0; br.stage_hi = 0;
            uint32_t np = 0, lg = 0, used = 0;
            int e = cz_fse_read_probs(br, t == 1 ? 8u : 9u, sh.a.t3.probs[t], &np, &lg, &used, 100);
            if (!e && used > left) e = CZ_E_DICT_TRUNCATED;
            if (!e) { cz_fse_build(cz_fse_table(t), sh.a.t3.probs[t], np, lg, sh.a.t3.counters[t], sh.b.c.llml, (uint32_t)t); sh.fse_log[t] = (uint8_t)lg; }
            bc.err = e; bc.d0 = used;
        }
        __syncthreads();
        err = cz_unii(bc.err); off += cz_uni((uint32_t)bc.d0);
        __syncthreads();
    }
    if (!err && len - off < 12) err = CZ_E_DICT_TRUNCATED;              /* (panic) :81-83 */
    if (!err) {
        for (uint32_t i = (uint32_t)LANE; i < 512; i += 64) { st->fse[0][i] = CZ_FSE_LL[i]; st->fse[2][i] = CZ_FSE_ML[i]; }
        for (uint32_t i = (uint32_t)LANE; i < 256; i += 64) st->fse[1][i] = CZ_FSE_OF[i];
        if (LANE == 0) {
            for (int k = 0; k < 3; k++) {
                cz_gcptr q = raw + off + 4 * k;
                st->hist[k] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);   /* :81-85 */
                st->fse_rle[k] = -1; st->fse_log[k] = sh.fse_log[k];
            }
            st->huf_max_bits = sh.huf_max_bits; st->dict_lag = 0;
        }
        off += 12;
    }
    if (LANE == 0) { result[0] = (uint64_t)(uint32_t)err; result[1] = off; result[2] = id; result[3] = magic; }
}

#endif /* !CZ_EXEC_ONLY */
#ifdef CZ_EXEC_ONLY
/* cz_execute_frames_kernel: the frames the pre-pass finished (chain records from cz_chain_kernel AND literals from
 * cz_huf_kernel) — block walk, record-driven execution of the sequences (sequence_execution.cairo:12-83), Raw / RLE blocks, checksum.
 * Same source as cz_decode_frames_kernel minus every decoder (no Huffman table, no FSE tables, no bit ring in LDS: 3 KB per
 * wave instead of 10.5 KB, and a register budget of its own).  Any other frame — and any frame that turns out to need a
 * decoder after all — is appended to args.fallback_list for cz_decode_frames_kernel. */
#ifndef CZ_EXEC_KERNEL
#define CZ_EXEC_KERNEL cz_execute_frames_kernel
#endif
extern "C" __global__ void __launch_bounds__(CZ_WG_THREADS, CZ_EXEC_WAVES) CZ_EXEC_KERNEL(cz_batch_args a) {
#ifdef CZ_PROFILE
    if (LANE == 0) for (int i = 0; i < CZ_P_COUNT; i++) sh.prof[i] = 0;
#endif
    if (cz_uni(a.scan_ctl[204]) == 0) return;                           /* every frame is CZ_PRE_DONE: nothing to walk (a shared work counter serves ~90 pulls per microsecond) */
    /* The EARLY launch (args.early == 1; the 4-waves build) runs beside the launch of the large blocks' chains, behind the small
       blocks' chains and the literal kernels: it takes exactly the frames cz_scan_kernel marked CZ_PRE_EARLY — no large block,
       so all of their pre-pass is behind a kernel boundary.  Later launches of such a batch (args.early == 2) may still find it
       at work: every frame is claimed with an atomic OR before it is executed. */
    const int early = a.early == 1u;
    if (early && cz_uni(a.scan_ctl[211]) == 0) return;
    if (!early && ::cz_exec_variant(a) != CZ_EXEC_WAVES) return;        /* the build with the other register budget runs this batch */
    const uint32_t total = a.n;
    const int wx_on = !early && ::cz_wx_side_by_side(a), wx_big = !early && ::cz_wx_big_only(a);
    /* side by side: cz_wexec_kernel needs whole CUs.  A wave that finds itself on an even CU gives that kernel's workgroups a few
       microseconds to count themselves in (scan_ctl[213] of args.wx_cus: dispatched first, as usual, they are there and nobody
       leaves); if they are not, this kernel was placed first and holds every CU: the waves on the even ones leave (cz_cu_side) */
#if !defined(CZ_EXP_NO_SIDE) && !defined(CZ_EMU)                       /* (the emulator runs the kernels one after the other: no CUs to share) */
    if (wx_on && ::cz_cu_side() == 1u) {
        uint32_t polls = 0;
        while (*(volatile uint32_t*)&a.scan_ctl[213] < a.wx_cus && polls < 8u) { __builtin_amdgcn_s_sleep(127); polls++; }
        if (*(volatile uint32_t*)&a.scan_ctl[213] < a.wx_cus) { if (LANE == 0) atomicAdd(&a.scan_ctl[215], 1u); return; }
        if (polls && LANE == 0) atomicAdd(&a.scan_ctl[216], 1u);
    }
#endif
    cz_init_llml();
    for (;;) {
        __syncthreads();
        if (LANE == 0) sh.frame_idx = atomicAdd(a.exec_counter, 1u);
        __syncthreads();
        const uint32_t fi = cz_uni(sh.frame_idx);
        if (fi >= total) break;
        const uint32_t f = a.frame_order ? cz_uni(a.frame_order[fi]) : fi;
        int err = CZX_FALLBACK;
        /* regular to its last block, everything listed, and cz_huf_kernel met nothing irregular; a frame whose chains
           cz_chain_kernel gave up on (first == 0 with sequences in it) comes back from cz_run_frame */
        const uint64_t first = cz_uni64(a.frame_first[f]), lfirst = cz_uni64(a.lit_first[f]);
        const uint32_t pre = cz_uni(a.frame_pre[f]);
        if (((pre & CZ_PRE_DONE) && lfirst != 0) || (pre & (CZ_PRE_WXDONE | CZ_PRE_CLAIMED | CZ_PRE_LISTED))) continue;   /* done by the pre-pass kernels / the other execute kernel's / handed back already */
        if (wx_big && (pre & CZ_PRE_WXBIG)) continue;                   /* cz_wexec_kernel's */
        if (early && !(pre & CZ_PRE_EARLY)) continue;                   /* waits for the large blocks' chains: the later launches */
        if (early || (a.early == 2u && !(wx_on && (pre & CZ_PRE_WXLIST)))) {
            __syncthreads();
            if (LANE == 0) {
                const uint32_t got = atomicOr(&a.frame_pre[f], CZ_PRE_CLAIMED);
                if (!(got & CZ_PRE_CLAIMED) && (got & CZ_PRE_WXLIST)) atomicAdd(&a.scan_ctl[208], 1u);   /* (listed frames claimed so far: see wx_leave) */
                sh.frame_idx = got;
            }
            __syncthreads();
            if (cz_uni(sh.frame_idx) & CZ_PRE_CLAIMED) continue;        /* the other launch has it */
        }
        if (wx_on && (pre & CZ_PRE_WXLIST)) {
            /* cz_wexec_kernel, which runs beside this kernel, may take this frame: whoever claims it first does it.  The last
               wx_leave listed frames are left to that kernel: a frame started here now would still be running on its one wave
               long after the other kernel has run out of frames. */
            __syncthreads();
            if (LANE == 0) {
                uint32_t got = CZ_PRE_CLAIMED;
                if (a.scan_ctl[206] - *(volatile uint32_t*)&a.scan_ctl[208] > a.wx_leave) { got = atomicOr(&a.frame_pre[f], CZ_PRE_CLAIMED); if (!(got & CZ_PRE_CLAIMED)) atomicAdd(&a.scan_ctl[208], 1u); }
                sh.frame_idx = got;
            }
            __syncthreads();
            if (cz_uni(sh.frame_idx) & CZ_PRE_CLAIMED) continue;
        }   /* all of it done by the pre-pass kernels, result record written by the scan (or taken back and listed by cz_huf_kernel) */
        if ((pre & CZ_PRE_REGULAR) && lfirst != 0) {
            CzFrameIO io;
            io.src = (cz_gcptr)(a.in_base + a.in_off[f]); io.src_len = a.in_len[f]; io.dst = (cz_gptr)(a.out_base + a.out_off[f]); io.dst_cap = a.out_cap[f];
            io.produced = 0; io.drained = 0; io.window = 0; io.parse_header = 1; io.has_checksum = 0;
            io.strategy = 0; io.strategy_n = 0; io.streaming = 0; io.verify = a.verify_checksum; io.dict = nullptr; io.dict_len = 0;
            io.pre_blocks = pre & CZ_PRE_COUNT;
            cz_state_reset();
            __syncthreads();
            CzLitPass lp; lp.arena = (cz_gptr)a.lit_arena; lp.on = 1; lp.cursor = lfirst;
            err = cz_run_frame(io, nullptr, nullptr, (CZ_GLOBAL cz_frame_result*)&a.results[f], (cz_gcptr64)a.chain_arena, first, lp);
            if (err != CZX_FALLBACK && LANE == 0 && a.results[f].status == 0 && !(a.results[f].flags & CZ_RESULT_FINISHED)) a.results[f].status = CZ_E_NOT_FINISHED;
        }
        if (err == CZX_FALLBACK && LANE == 0) ::cz_list_fallback(a, f);
#ifdef CZ_PROFILE
        if (LANE == 0 && a.prof) for (int i = 0; i < CZ_P_COUNT; i++) { atomicAdd(&a.prof[i], sh.prof[i]); sh.prof[i] = 0; }
#endif
    }
}
#else
/* Persistent grid: every workgroup (one wavefront) pulls frames off a shared counter. */
extern "C" __global__ void __launch_bounds__(CZ_WG_THREADS, CZ_MAIN_WAVES) cz_decode_frames_kernel(cz_batch_args a) {
    if (a.fallback_list && cz_uni(*a.fallback_count) == 0) return;      /* behind cz_execute_frames_kernel, and it left nothing */
    cz_init_llml();
    cz_gptr lit_scratch = (cz_gptr)(a.lit_scratch + (uint64_t)blockIdx.x * a.lit_scratch_stride);
#ifdef CZ_PROFILE
    if (LANE == 0) for (int i = 0; i < CZ_P_COUNT; i++) sh.prof[i] = 0;
#endif
    for (;;) {
        __syncthreads();
        if (LANE == 0) sh.frame_idx = atomicAdd(a.work_counter, 1u);
        __syncthreads();
        const uint32_t fi = cz_uni(sh.frame_idx);
        /* behind cz_execute_frames_kernel: only the frames it left (args.fallback_list, in the order it met them) */
        if (fi >= (a.fallback_list ? (cz_uni(*a.fallback_count) < a.n ? cz_uni(*a.fallback_count) : a.n) : a.n)) break;
        const uint32_t f = a.fallback_list ? cz_uni(a.fallback_list[fi]) : (a.frame_order ? cz_uni(a.frame_order[fi]) : fi);   /* the pre-pass sorted the frames: longest first */
        CzFrameIO io;
        if (a.tasks) {
            const cz_device_task t = a.tasks[f];
            io.src = (cz_gcptr)t.src; io.src_len = t.src_len; io.dst = (cz_gptr)t.dst; io.dst_cap = t.dst_cap; io.produced = t.produced;
            io.drained = t.drained; io.window = t.window_size; io.parse_header = 0; io.has_checksum = t.has_checksum;
            io.strategy = t.strategy; io.strategy_n = t.strategy_n; io.streaming = t.streaming; io.verify = 0;
            io.dict = (cz_gcptr)t.dict; io.dict_len = t.dict_len; io.pre_blocks = 0;
            /* restore carried state (the Huffman table stays in t.state->huf until a Treeless block asks for it) */
            cz_device_frame_state* gs = t.state;
            for (uint32_t i = (uint32_t)LANE; i < 512; i += 64) { CZ_FSE_LL[i] = gs->fse[0][i]; CZ_FSE_ML[i] = gs->fse[2][i]; }
            for (uint32_t i = (uint32_t)LANE; i < 256; i += 64) CZ_FSE_OF[i] = gs->fse[1][i];
            if (LANE == 0) {
                for (int k = 0; k < 3; k++) { sh.hist[k] = gs->hist[k]; sh.fse_rle[k] = gs->fse_rle[k]; sh.fse_log[k] = gs->fse_log[k]; }
                sh.huf_max_bits = gs->huf_max_bits;
                sh.dict_lag[0] = (uint32_t)gs->dict_lag; sh.dict_lag[1] = (uint32_t)(gs->dict_lag >> 32);
            }
            __syncthreads();
            CzLitPass nolit; nolit.arena = nullptr; nolit.cursor = 0; nolit.on = 0;
            cz_run_frame(io, lit_scratch, (cz_gptr16)gs->huf, (CZ_GLOBAL cz_frame_result*)&a.results[f], nullptr, 0, nolit);
            for (uint32_t i = (uint32_t)LANE; i < 512; i += 64) { gs->fse[0][i] = CZ_FSE_LL[i]; gs->fse[2][i] = CZ_FSE_ML[i]; }
            for (uint32_t i = (uint32_t)LANE; i < 256; i += 64) gs->fse[1][i] = CZ_FSE_OF[i];
            if (LANE == 0) {
                for (int k = 0; k < 3; k++) { gs->hist[k] = sh.hist[k]; gs->fse_rle[k] = sh.fse_rle[k]; gs->fse_log[k] = sh.fse_log[k]; }
                gs->huf_max_bits = sh.huf_max_bits;
                gs->dict_lag = ((uint64_t)sh.dict_lag[1] << 32) | sh.dict_lag[0];
            }
        } else {
            io.src = (cz_gcptr)(a.in_base + a.in_off[f]); io.src_len = a.in_len[f]; io.dst = (cz_gptr)(a.out_base + a.out_off[f]); io.dst_cap = a.out_cap[f];
            io.produced = 0; io.drained = 0; io.window = 0; io.parse_header = 1; io.has_checksum = 0;
            io.strategy = 0; io.strategy_n = 0; io.streaming = 0; io.verify = a.verify_checksum; io.dict = (cz_gcptr)a.dict; io.dict_len = a.dict_len;
            cz_state_reset();
            if (a.dict_state) {
                /* every frame of the batch starts as DecoderScratch::init_from_dict leaves a workspace (scratch.cairo:60-65):
                   the dictionary's tables and repeat offsets; its Huffman table goes to this workgroup's carried-table slot */
                CZ_GLOBAL const cz_device_frame_state* ds = (CZ_GLOBAL const cz_device_frame_state*)a.dict_state;
                __syncthreads();
                for (uint32_t i = (uint32_t)LANE; i < 512; i += 64) { CZ_FSE_LL[i] = ds->fse[0][i]; CZ_FSE_ML[i] = ds->fse[2][i]; }
                for (uint32_t i = (uint32_t)LANE; i < 256; i += 64) CZ_FSE_OF[i] = ds->fse[1][i];
                for (uint32_t i = (uint32_t)LANE; i < 1024; i += 64) ((CZ_GLOBAL uint32_t*)(lit_scratch + CZ_LIT_SCRATCH_BYTES))[i] = ((CZ_GLOBAL const uint32_t*)ds->huf)[i];
                if (LANE == 0) {
                    for (int k = 0; k < 3; k++) { sh.hist[k] = ds->hist[k]; sh.fse_rle[k] = ds->fse_rle[k]; sh.fse_log[k] = ds->fse_log[k]; }
                    sh.huf_max_bits = ds->huf_max_bits;
                }
            }
            __syncthreads();
            /* what the pre-pass left for this frame: literal nodes / literals done (cz_huf_kernel) and leading blocks already in place */
            CzLitPass lp; lp.arena = (cz_gptr)a.lit_arena;
            lp.cursor = a.lit_arena ? cz_uni64(a.lit_first[f]) : 0; lp.on = lp.cursor != 0;
            io.pre_blocks = lp.on ? cz_uni(a.frame_pre[f]) & CZ_PRE_COUNT : 0u;
            cz_run_frame(io, lit_scratch, (cz_gptr16)(lit_scratch + CZ_LIT_SCRATCH_BYTES), (CZ_GLOBAL cz_frame_result*)&a.results[f], (cz_gcptr64)a.chain_arena,
                         a.chain_arena ? cz_uni64(a.frame_first[f]) : 0, lp);
#ifdef CZ_PROFILE
            if (LANE == 0 && a.prof) for (int i = 0; i < CZ_P_COUNT; i++) { atomicAdd(&a.prof[i], sh.prof[i]); sh.prof[i] = 0; }
#endif
            if (LANE == 0 && a.results[f].status == 0 && !(a.results[f].flags & CZ_RESULT_FINISHED)) a.results[f].status = CZ_E_NOT_FINISHED;
        }
    }
}
#endif /* CZ_EXEC_ONLY */
