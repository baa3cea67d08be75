/*
 * czstd_wexec.hip — cz_wexec_kernel: sequence execution with SEVERAL waves per frame and the frame's window in LDS.
 *
 * cz_execute_frames_kernel gives a frame to one wave and reads match sources from the frame's output in HBM: with 4 096 frames in
 * flight that is a 512 MB footprint of random 32-byte reads.  Here a frame is the work of one WORKGROUP of up to 16
 * waves, one workgroup per CU, and the frame's output is assembled in a 128 KiB window in LDS: match sources never leave the CU,
 * and the output goes to HBM once, with aligned 16-byte stores, when a block is finished.
 *
 * What the reference carries from one sequence to the next (sequence_execution.cairo:12-129, scratch.cairo:11-19) is the output
 * position, the literal cursor and the three-entry offset history.  All three compose associatively, so a block's records are cut
 * into CHUNKS of 64 sequences that different waves work on at once (chunk c belongs to wave c mod W):
 *   1  records -> (ll, ml, offset_value) per lane; prefix sums of ll and ll + ml; the chunk's history transform by a DPP scan over
 *      packed transforms (cz_history's, with SYMBOLIC results: "what slot j held when the chunk began, plus delta");
 *   2  the chunk waits for the state BEFORE it — published by chunk c - 1 in an LDS slot —, adds its own sums, resolves its
 *      summary against the incoming history and publishes the state before chunk c + 1.  Nothing else is serial: the chain of
 *      publications runs ahead of the data movement;
 *   3  literals go from the literal buffer (HBM, read once) to their place in the window;
 *   4  matches copy window -> window.  A match may go once every byte of its source is final.  That is kept per byte: a bitmap
 *      of the window in LDS, a bit set (ds_or) when the byte's literal or match copy has been written.  Below a static HORIZON —
 *      the output position of the chunk this wave worked on two turns ago: every chunk before that one is complete, because a
 *      wave takes its chunks in order — nothing is looked up; above it a lane tests the words of the bitmap that cover its
 *      source.  The oldest chunk in flight never waits for another one, so the pipeline cannot lock up.
 * A wave keeps TWO chunks in flight (step 1 of the next chunk is issued before the data movement of the current one), and step 2
 * is a decoupled look-back over the last 64 entries rather than a wait for the predecessor (entry flags: sums there / outgoing
 * history there / state BEHIND the chunk there).
 * Per block: barrier, tail literals (sequence_execution.cairo:72-78), the block's bytes window -> HBM.  The window positions are
 * block-relative; a block whose output exceeds the window is done in passes (the first sequence that does not fit sets
 * ctl.reset_at; flush; the next pass starts there; a chunk that alone exceeds the window goes through wx_slow_chunk between two
 * passes), and sources in earlier blocks are read from the frame's output in HBM.
 *
 * The kernel is a pure accelerator, like cz_chain_kernel: it takes frames off the list cz_scan_kernel made for it (regular to the
 * last block, everything pre-passed, enough sequences to be worth a workgroup), claims each with an atomic OR on frame_pre[f]
 * (CZ_PRE_CLAIMED) and marks those it finished (CZ_PRE_WXDONE, result record written).  cz_execute_frames_kernel runs SIDE BY SIDE
 * with it on the other CUs, claims frames the same way and leaves the last few listed frames per workgroup to this kernel.  The
 * whole kernel returns at once unless the batch's offset codes are mostly far ones (cz_wx_side_by_side: the sums cz_chain_kernel
 * left in chain_top): near-offset frames are bound by their chain of dependent matches and gain nothing here (profiles/r4/NOTES.md)
 * — except the few LARGE frames of a batch that fills the chip (cz_wx_big_only: CZ_PRE_WXBIG, marked by the scan): the batch ends
 * when they do, and here each has a CU and its window to itself instead of a wave among 4 096; then this kernel does exactly the
 * marked frames and cz_execute_frames_kernel leaves them alone.
 * On ANY irregularity — a check of execute_sequences that fails, a frame the pre-pass kernels took back — the frame goes on
 * fallback_list as it was, and cz_decode_frames_kernel does it in the reference's order of detection.  Nothing here reports errors.
 */
#define WX_WAVES 16u
#define WX_THREADS (64u * WX_WAVES)
#define WX_RING CZ_WX_RING                   /* the window, bytes */
#define WX_NS 64u                            /* look-back entries, one per chunk in flight and the 48 before (an entry is not reused before its readers are done with it) */
#define WX_F_AGG 1u                          /* entry flags: the chunk's sums are there / its outgoing history / the state BEHIND the chunk */
#define WX_F_HOK 2u
#define WX_F_INCL 4u
#define WX_INF 0xFFFFFFFFu
#define WX_REL 0x80000000u                   /* symbolic history value: bits 30:29 = incoming slot, bits 15:0 = 0x8000 + delta */
#define WX_COOP_LEN 96u                      /* literal runs / matches longer than this are copied by the whole wave */

/* LDS words another wave writes: volatile accesses, and fences that keep the compiler from moving the window accesses across them.
 * On the device the LDS executes one wave's instructions in order, so a flag written after the data is seen after the data; the
 * CPU emulator (tests/emu) runs lanes as threads and needs real fences. */
#ifdef CZ_EMU
#define WX_FENCE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define WX_PAUSE() sched_yield()
#include <stdio.h>
#define WX_SPIN_GUARD(cnt, ...) do { if (++(cnt) == 20000u) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)
#define WX_DBG(...) do { if (getenv("EMU_WX_DBG")) { fprintf(stderr, __VA_ARGS__); fflush(stderr); } } while (0)
#define WX_AT(n) do { emu_site[threadIdx.x] = (void*)(uintptr_t)(0x1000 + (n)); } while (0)   /* where a lane is, for the emulator's watchdog */
/* a word another wave may change while this wave reads it, made wave-uniform: on the device v_readfirstlane; the emulator's
   readfirstlane is the identity (its callers pass uniform values), so here the lanes really take lane 0's copy — else a wave's
   lanes part ways on ctl.reset_at / ctl.err and its barriers no longer pair up */
#define WX_UNI(v) ((uint32_t)__shfl((int)(v), 0))
#else
#define WX_UNI(v) cz_uni(v)
#define WX_AT(n) do { } while (0)
#define WX_FENCE() asm volatile("" ::: "memory")
#define WX_PAUSE() __builtin_amdgcn_s_sleep(1)
#define WX_SPIN_GUARD(cnt, ...) do { } while (0)
#define WX_DBG(...) do { } while (0)
#endif
/* Every wait on another wave is BOUNDED: the two polling loops (a match that waits for its source bytes, a look-back that waits
 * for the entries before its chunk) count their polls, and past WX_SPIN_LIMIT the wave sets ctl.err — the frame is given up and
 * goes to cz_decode_frames_kernel like any other frame this kernel cannot finish, and every other wave of the workgroup leaves its
 * own wait at its next poll.  A correct protocol never gets near the bound (a poll is >= 64 clocks of s_sleep plus the LDS reads:
 * the limit is >= 30 ms of waiting, a whole frame takes 0.1-5 ms); it is there because the input is untrusted bytes and a hung
 * wave is a hung GPU.  The emulator's limit is small, so that the test that forces it (CZ_DEBUG_WX_POISON) runs in seconds. */
#ifdef CZ_EMU
#define WX_SPIN_LIMIT 3000u
#define WX_CHAIN_WAIT_POLLS 2000u
#else
#define WX_SPIN_LIMIT (1u << 20)
#define WX_CHAIN_WAIT_POLLS (1u << 15)      /* x >= 4 096 clocks of s_sleep: >= 50 ms for the chain of ONE block (131 072 sequences take 12 ms) */
#endif

/* diagnostic build only (-DCZ_PROFILE): per-wave s_memtime sums per phase, added to args.prof[40..49] when the kernel ends */
#ifdef CZ_PROFILE
#define WX_PROF_T0() do { wxt_ = __builtin_amdgcn_s_memtime(); } while (0)
#define WX_PROF_ACC(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); wxp[i] += n_ - wxt_; wxt_ = n_; } while (0)
#define WX_PROF_CNT(i) do { wxp[i] += 1; } while (0)
#else
#define WX_PROF_T0() do { } while (0)
#ifdef WX_ASM_MARKS
#define WX_PROF_ACC(i) asm volatile("; WX_MARK " #i)
#else
#define WX_PROF_ACC(i) do { } while (0)
#endif
#define WX_PROF_CNT(i) do { } while (0)
#endif
struct WxCtl {
    uint32_t llml[96];                                         /* [0..35] LL base | bits << 24, [40..92] ML */
    __attribute__((aligned(16))) uint8_t maps[CZ_CHAIN_MAP_BYTES];   /* state -> code: LL 512, ML 512, OF 256 (kept over Repeat-mode blocks) */
    __attribute__((aligned(16))) uint32_t slot[WX_NS][8];      /* chunk c at [c % WX_NS]: (c + 1) << 3 | flags, its sum of ll + ml, of ll, position behind it | literal cursor behind it, history behind it */
    uint32_t fin[WX_RING / 32u + 4u];                          /* one bit per byte of the window: final (set by the sequence that wrote it) */
    uint32_t err;                                              /* some wave met something irregular: the frame is left to cz_execute_frames_kernel */
    uint32_t reset_at;                                         /* first chunk of the block whose output does not fit the window any more (WX_INF: none) */
    uint32_t slow_at;                                          /* ... and, when it is the same chunk, it would not fit an EMPTY window either: wx_slow_chunk does it (WX_INF: none) */
    uint32_t slow_ok, slow_P, slow_L;                          /* wx_slow_chunk: its checks passed; positions behind the chunk */
    /* written by thread 0 between barriers */
    uint32_t fidx;
    uint32_t go;                                               /* 0 end of frame (ok), 1 block follows, 2 give the frame up */
    uint32_t btype, bsize, blt, bregen, bnseq, bpredone, mapflags, lit_rle, lit_byte, lit_len;
    uint32_t P, blocks, hist[3];
    uint32_t src_lo, src_hi, lit_lo, lit_hi, rec_lo, rec_hi, maps_lo, maps_hi, bits_lo, bits_hi;
    uint8_t dump[64 * WX_WAVES + 16];                          /* where a lane's byte goes when it has none to write */
};
#define WX_LDS_BYTES (WX_RING + (uint32_t)sizeof(WxCtl))

/* (the pointers are cast to the LDS address space by hand: the compiler does not infer it for volatile accesses and would emit
   flat_* instructions, whose waits also cover every global load in flight) */
#if defined(__HIP_DEVICE_COMPILE__)
#define WX_LDS __attribute__((address_space(3)))
#else
#define WX_LDS
#endif
__device__ static inline uint32_t wx_ld(const uint32_t* p) { return *(const volatile WX_LDS uint32_t*)p; }
__device__ static inline void wx_st(uint32_t* p, uint32_t v) { *(volatile WX_LDS uint32_t*)p = v; }
__device__ static inline uint64_t wx_ptr(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
/* a look-back entry: the word with the flags is read first (device: it is in the first of two 16-byte LDS reads, which execute in order) */
__device__ static inline void wx_read_entry(const uint32_t* e, uint32_t* w) {
#ifdef CZ_EMU
    w[0] = wx_ld(&e[0]); WX_FENCE();
    for (int k = 1; k < 8; k++) w[k] = wx_ld(&e[k]);
#else
    typedef uint32_t wx_v4 __attribute__((ext_vector_type(4)));
    const wx_v4 a = *(const volatile WX_LDS wx_v4*)e; const wx_v4 b = *(const volatile WX_LDS wx_v4*)(e + 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
#endif
}

/* unaligned window accesses (gfx950 LDS takes them; the compiler emits ds_read_b32 / b64 for these) */
__device__ static inline uint32_t wx_r32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ static inline uint64_t wx_r64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ static inline void wx_w16(uint8_t* p, uint32_t v) { const uint16_t h = (uint16_t)v; __builtin_memcpy(p, &h, 2); }
__device__ static inline void wx_w32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ static inline void wx_w64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
/* exactly n <= 8 bytes of v */
__device__ static inline void wx_wn(uint8_t* p, uint64_t v, uint32_t n) {
    if (n >= 8u) { wx_w64(p, v); return; }
    if (n & 4u) { wx_w32(p, (uint32_t)v); p += 4; v >>= 32; }
    if (n & 2u) { wx_w16(p, (uint32_t)v); p += 2; v >>= 16; }
    if (n & 1u) *p = (uint8_t)v;
}
/* up to 8 bytes from global memory: one load when it stays inside the buffer (`whole`), else byte by byte */
__device__ static inline uint64_t wx_g64(cz_gcptr s, uint32_t n, int whole) {
    if (whole) return cz_ldu64(s);
    uint64_t v = 0;
    for (uint32_t b = 0; b < 8; b++) if (b < n) v |= (uint64_t)s[b] << (8 * b);
    return v;
}

/* ---- copies by the whole workgroup (blocks without sequences, tails, the flush) */
/* global -> window */
__device__ static inline void wx_wg_g2w(uint8_t* ring, uint32_t at, cz_gcptr src, uint32_t n, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = 16u * tid; i < n; i += 16u * nthreads) {
        if (i + 16u <= n) { const uint4 v = cz_ldu128(src + i); __builtin_memcpy(ring + at + i, &v, 16); }
        else for (uint32_t j = i; j < n; j++) ring[at + j] = src[j];
    }
}
__device__ static inline void wx_wg_fill(uint8_t* ring, uint32_t at, uint32_t byte, uint32_t n, uint32_t tid, uint32_t nthreads) {
    const uint32_t w = 0x01010101u * byte; const uint4 v = uint4{w, w, w, w};
    for (uint32_t i = 16u * tid; i < n; i += 16u * nthreads) {
        if (i + 16u <= n) __builtin_memcpy(ring + at + i, &v, 16);
        else for (uint32_t j = i; j < n; j++) ring[at + j] = (uint8_t)byte;
    }
}
/* window -> global, aligned 16-byte stores once the destination is aligned */
__device__ static inline void wx_wg_flush(const uint8_t* ring, uint32_t at, cz_gptr dst, uint32_t n, uint32_t tid, uint32_t nthreads) {
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > n) head = n;
    if (tid < head) dst[tid] = ring[at + tid];
    const uint32_t body = (n - head) >> 4;
    for (uint32_t i = tid; i < body; i += nthreads) { uint4 v; __builtin_memcpy(&v, ring + at + head + 16u * i, 16); *(cz_gptr4)(dst + head + 16u * i) = v; }
    const uint32_t done = head + 16u * body;
    if (tid < n - done) dst[done + tid] = ring[at + done + tid];
}

/* ---- copies by one wave (long literal runs and long matches of a chunk) */
__device__ static inline void wx_wave_g2w(uint8_t* ring, uint32_t at, cz_gcptr src, uint32_t n) {
    for (uint32_t i = 8u * (uint32_t)LANE; i < n; i += 512u) {
        const uint32_t m = n - i < 8u ? n - i : 8u;
        wx_wn(ring + at + i, wx_g64(src + i, m, m == 8u), m);
    }
}
__device__ static inline void wx_wave_fill(uint8_t* ring, uint32_t at, uint32_t byte, uint32_t n) {
    const uint64_t v = 0x0101010101010101ull * byte;
    for (uint32_t i = 8u * (uint32_t)LANE; i < n; i += 512u) wx_wn(ring + at + i, v, n - i < 8u ? n - i : 8u);
}
/* window -> window, n bytes to d from d - off: the forward byte copy of decode_buffer.cairo:95-127.  With off < n the output is
   periodic from d - off on, so any earlier multiple of off is as good a distance: the distance doubles until it covers a step. */
__device__ static inline void wx_wave_w2w(uint8_t* ring, uint32_t d, uint32_t off, uint32_t n) {
    uint32_t copied = 0, dist = off;
    while (copied < n) {
        while (dist < 512u && 2u * dist <= off + copied) dist += dist;
        uint32_t step = n - copied < dist ? n - copied : dist;
        if (step > 512u) step = 512u;
        const uint32_t i = 8u * (uint32_t)LANE;
        uint64_t v = 0; uint32_t m = 0;
        if (i < step) { m = step - i < 8u ? step - i : 8u; v = wx_r64(ring + d + copied - dist + i); }   /* (reads at most 7 bytes beyond the source; only m are used) */
        cz_wave_sync();                                                 /* every lane has read before any lane writes */
        if (m) wx_wn(ring + d + copied + i, v, m);
        cz_wave_sync();
        copied += step;
    }
}

/* ---- one lane: n bytes to d from d - off, any n (the per-lane loop behind the short forms) */
__device__ static inline void wx_lane_w2w(uint8_t* ring, uint32_t d, uint32_t off, uint32_t n) {
    uint32_t copied = 0, dist = off;
    while (copied < n) {
        while (dist < 8u && 2u * dist <= off + copied) dist += dist;
        uint32_t step = n - copied < dist ? n - copied : dist;
        if (step > 8u) step = 8u;
        wx_wn(ring + d + copied, wx_r64(ring + d + copied - dist), step);
        copied += step;
    }
}

/* ---- the `final` bitmap: bit p = byte p of the window has its final value */
#ifdef CZ_EMU
#define WX_OR(p, v) ((void)__atomic_fetch_or((p), (v), __ATOMIC_SEQ_CST))
static inline void wx_emu_min(uint32_t* p, uint32_t v) { uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v < o && !__atomic_compare_exchange_n(p, &o, v, 0, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { } }
#define WX_MIN(p, v) wx_emu_min((p), (v))
#else
#define WX_OR(p, v) ((void)__hip_atomic_fetch_or((WX_LDS uint32_t*)(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define WX_MIN(p, v) ((void)__hip_atomic_fetch_min((WX_LDS uint32_t*)(p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#endif
/* the bits of [p, p + n) that lie in 32-bit word `word` of the bitmap */
__device__ static inline uint32_t wx_word_mask(uint32_t p, uint32_t n, uint32_t word) {
    const uint32_t lo = word << 5, hi = lo + 32u, a = p > lo ? p : lo, e = p + n < hi ? p + n : hi;
    if (a >= e) return 0u;
    const uint32_t len = e - a;
    return (len >= 32u ? 0xFFFFFFFFu : ((1u << len) - 1u)) << (a & 31u);
}
/* n <= 32: two words at most */
__device__ static inline void wx_mark32(uint32_t* fin, uint32_t p, uint32_t n) {
    const uint64_t m = ((1ull << n) - 1ull) << (p & 31u);
    WX_OR(&fin[p >> 5], (uint32_t)m); WX_OR(&fin[(p >> 5) + 1u], (uint32_t)(m >> 32));
}
__device__ static inline int wx_final32(const uint32_t* fin, uint32_t p, uint32_t n) {
    const uint64_t m = ((1ull << n) - 1ull) << (p & 31u);
    const uint32_t w0 = wx_ld(&fin[p >> 5]), w1 = wx_ld(&fin[(p >> 5) + 1u]);
    return (((uint32_t)m & ~w0) | ((uint32_t)(m >> 32) & ~w1)) == 0u;
}
/* any n, one lane */
__device__ static inline void wx_mark(uint32_t* fin, uint32_t p, uint32_t n) {
    for (uint32_t w = p >> 5; (w << 5) < p + n; w++) WX_OR(&fin[w], wx_word_mask(p, n, w));
}
__device__ static inline int wx_final(const uint32_t* fin, uint32_t p, uint32_t n) {
    for (uint32_t w = p >> 5; (w << 5) < p + n; w++) { const uint32_t m = wx_word_mask(p, n, w); if (m & ~wx_ld(&fin[w])) return 0; }
    return 1;
}
/* any n, the whole wave */
__device__ static inline void wx_wave_mark(uint32_t* fin, uint32_t p, uint32_t n) {
    for (uint32_t w = (p >> 5) + (uint32_t)LANE; (w << 5) < p + n; w += 64u) WX_OR(&fin[w], wx_word_mask(p, n, w));
}

/* records -> values (cz_rec_values with this kernel's tables).  WIDE = 0: no record of the chunk has more than 32 extra bits */
template <int WIDE>
__device__ static inline uint32_t wx_rec_values(const WxCtl& ctl, uint64_t r, cz_gcptr bits, uint32_t& ll, uint32_t& ml) {
    const uint8_t* mapll = ctl.maps; const uint8_t* mapml = mapll + 512; const uint8_t* mapof = mapll + 1024;
    const uint32_t xt = (uint32_t)r, st = (uint32_t)(r >> 32);
    const uint32_t oc = mapof[(st >> 18) & 255];
    const uint32_t tl = ctl.llml[mapll[st & 511]], tm = ctl.llml[40 + mapml[(st >> 9) & 511]];
    const uint32_t mx = tm >> 24, lx = tl >> 24;
    uint32_t ov;
    if (!WIDE || !(st & CZC_REC_WIDE)) {                                /* <= 32 extra bits: the top of the record's low word */
        ov = (1u << oc) + __builtin_amdgcn_ubfe(xt, 32 - oc, oc);       /* sequence_section_decoder.cairo:243 */
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

/* symbolic value of what a history slot holds, from a transform byte: pushed by a lane of this chunk, or an incoming slot */
__device__ static inline uint32_t wx_resolve(uint32_t s, uint32_t h0, uint32_t h1, uint32_t h2) {
    const uint32_t base = cz_pick3((s >> 29) & 3u, h0, h1, h2);
    return (s & WX_REL) ? base + (s & 0xFFFFu) - 0x8000u : s;
}

/* Step 1 of a chunk: values, the two prefix sums, the chunk's history transform (everything that does not depend on the chunks before) */
struct WxStep1 { uint32_t ll, ml, orel, lrel, asym, sum_ll, sum_tot, o0, o1, o2; int active, bad, habs; };
__device__ static inline WxStep1 wx_step1(const WxCtl& ctl, uint64_t r, cz_gcptr bits, uint32_t nseq, uint32_t y) {
    const uint32_t lane = (uint32_t)LANE;
    WxStep1 s1;
    const uint32_t cnt = nseq - 64u * y < 64u ? nseq - 64u * y : 64u;
    const int active = lane < cnt;
    uint32_t ll = 0, ml = 0, ov = 4;
    if (!__ballot((uint32_t)(r >> 32) & CZC_REC_WIDE)) { const uint32_t v = wx_rec_values<0>(ctl, r, bits, ll, ml); if (active) ov = v; else { ll = 0; ml = 0; } }
    else if (active) ov = wx_rec_values<1>(ctl, r, bits, ll, ml);
    s1.bad = active && ov >= 0x40000000u;                        /* (also keeps pushed values clear of WX_REL) */
    const uint32_t tot = ll + ml;
    if (!__ballot((ll | ml) >= 512u)) {                         /* both prefix sums in one scan */
        const uint32_t pk = ll | (tot << 16), incl = cz_wave_incl_scan(pk), sums = cz_readlane(incl, 63), excl = incl - pk;
        s1.sum_ll = sums & 0xFFFFu; s1.sum_tot = sums >> 16; s1.lrel = excl & 0xFFFFu; s1.orel = excl >> 16;
    } else {
        const uint32_t il = cz_wave_incl_scan(ll), it = cz_wave_incl_scan(tot);
        s1.sum_ll = cz_readlane(il, 63); s1.sum_tot = cz_readlane(it, 63); s1.lrel = il - ll; s1.orel = it - tot;
    }
    /* history (sequence_execution.cairo:85-129): cz_history's packed transforms; pushed values stay symbolic where they are
       "offset_value 3 with no literals" = what slot 0 held before the lane, minus one */
    uint32_t T, M;
    const int dec = active && ov == 3 && ll == 0;
    {
        const uint32_t kind = !active ? 0u : (ov > 3 ? 3u : ov - (ll > 0 ? 1u : 0u));
        const uint32_t t01 = (kind & 1u) ? 0x00020001u : CZ_T_ID, t23 = (kind & 1u) ? (0x00010080u | lane) : 0x00010002u;
        T = (kind & 2u) ? t23 : t01;
        M = kind == 3u ? 0xFFu : 0u;
    }
#define WX_HT_STEP(CTRL, RM) do { const uint32_t pT = cz_dpp<CTRL, RM>(CZ_T_ID, T), pM = cz_dpp<CTRL, RM>(0u, M); \
    const uint32_t R = __builtin_amdgcn_perm(T, pT, T); const uint32_t Mn = __builtin_amdgcn_perm(M, pM, T); \
    T = (T & M) | (R & ~M); M = Mn; } while (0)
    WX_HT_STEP(CZ_DPP_SHR1, 0xF); WX_HT_STEP(CZ_DPP_SHR2, 0xF); WX_HT_STEP(CZ_DPP_SHR4, 0xF); WX_HT_STEP(CZ_DPP_SHR8, 0xF);
    WX_HT_STEP(CZ_DPP_BCAST15, 0xA); WX_HT_STEP(CZ_DPP_BCAST31, 0xC);
#undef WX_HT_STEP
    uint32_t pv = ov - 3u;
    const unsigned long long dm = __ballot(dec);
    if (dm) {
        const uint32_t eT = cz_dpp<CZ_DPP_WAVE_SHR1, 0xF>(CZ_T_ID, T);  /* transform of everything before the lane */
        for (unsigned long long m = dm; m; m &= m - 1) {
            const int j = cz_unii(__ffsll((long long)m) - 1);
            const uint32_t bT = cz_readlane(eT, j) & 0xFFu;
            const uint32_t before = (bT & 0x80u) ? cz_readlane(pv, cz_unii((int)(bT & 63u))) : (WX_REL | ((bT & 3u) << 29) | 0x8000u);
            if ((int)lane == j) pv = before - 1u;
        }
    }
    const uint32_t pushed = __shfl(pv, (int)(T & 63u));         /* every lane takes part */
    s1.asym = (T & 0x80u) ? pushed : (WX_REL | ((T & 3u) << 29) | 0x8000u);
    const uint32_t fT = cz_readlane(T, 63);
    uint32_t osym[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t tk = (fT >> (8 * k)) & 0xFFu;
        const uint32_t pk_ = cz_readlane(pv, cz_unii((int)(tk & 63u)));
        osym[k] = (tk & 0x80u) ? pk_ : (WX_REL | ((tk & 3u) << 29) | 0x8000u);
    }
    s1.ll = ll; s1.ml = ml; s1.active = active;
    s1.o0 = cz_uni(osym[0]); s1.o1 = cz_uni(osym[1]); s1.o2 = cz_uni(osym[2]);
    s1.habs = !((s1.o0 | s1.o1 | s1.o2) & WX_REL);
    return s1;
}

/* Matches of one chunk (step 4), in rounds: a match goes once every byte of its source is final.  Below `horizon` — the end of the
 * chunk this wave worked on two turns ago — everything is: a chunk passes its look-back only when all chunks before it have
 * published their sums, i.e. when their waves have finished the data phase two chunks before those.  Above it the `final` bitmap
 * says so byte by byte (a sequence sets the bits of its literals and its match when the match is written), which also orders the
 * matches of the chunk itself.  first = 1: one round, returns the lanes that have to wait; first = 0: rounds until `todo` is done. */
struct WxData { uint32_t ll, ml, opos, off; uint64_t lw; uint32_t lpos; int active; };   /* a chunk ready for its data phase, per lane */
__device__ static inline unsigned long long wx_match_rounds(WxCtl& ctl, uint8_t* rw, cz_gcptr out, uint32_t rbase, uint32_t cap, const WxData& x, uint32_t horizon, uint8_t* dump, const int first,
                                                            unsigned long long* wxp, unsigned long long todo = 0) {
    const uint32_t lane = (uint32_t)LANE;
    const uint32_t ll = x.ll, ml = x.ml, opos = x.opos, off = x.off, d = opos + ll, tot = ll + ml;
    /* A source that begins before the block (position < rbase) is in the frame's output in HBM: the first n1 bytes of the match
       come from there (final since the barrier behind the earlier block), the rest, if any, from the start of the window on.
       The bytes to wait for: the window part of the source from `horizon` on (below it everything is final), clipped to the start of
       the sequence's own literals (they are in place). */
    const uint32_t src = d - off, n1 = src < rbase ? (rbase - src < ml ? rbase - src : ml) : 0u;
    const uint32_t send_ = off < ml ? d : src + ml, wend = send_ < opos ? send_ : opos, wsrc = src > horizon ? src : horizon,
                   slen = wsrc >= wend ? 0u : wend - wsrc;
    int undone = first ? x.active : (int)((todo >> lane) & 1ull);
    const int smallseq = !__ballot(tot > 32u);
    const int small4 = !__ballot(ml > 4u || n1 != 0u);
    const unsigned long long longm = __ballot(undone && ml > WX_COOP_LEN), hugem = __ballot(undone && tot > 512u);
    uint32_t spins = 0;
    for (;;) {
        if (!__ballot(undone)) break;
        WX_PROF_CNT(7);
        int ready = undone && slen == 0u;
        if (__ballot(undone && !ready)) {
            if (undone && !ready) ready = smallseq ? wx_final32(ctl.fin, wsrc - rbase, slen) : wx_final(ctl.fin, wsrc - rbase, slen);
            WX_FENCE();
        }
        const unsigned long long rm = __ballot(ready);
        if (!rm) {                                                      /* every match left waits for another wave (or the frame has been given up) */
            if (first || WX_UNI(wx_ld(&ctl.err))) break;
            if (++spins > WX_SPIN_LIMIT) { if (lane == 0) wx_st(&ctl.err, 1u); break; }   /* bounded: give the frame up */
            WX_PAUSE(); WX_PROF_CNT(8);
            WX_SPIN_GUARD(spins, "WX SPIN match: lane %u undone %d opos %u ll %u ml %u off %u src %u slen %u horizon %u rbase %u\n", lane, undone, opos, ll, ml, off, src, slen, horizon, rbase);
            continue;
        }
        if (small4) {
            if (ready) {
                uint32_t w = wx_r32(rw + src);
                if (off < 4u) w = off == 1u ? (w & 0xFFu) * 0x01010101u : (off == 2u ? (w & 0xFFFFu) * 0x00010001u : ((w & 0xFFFFFFu) | (w << 24)));
                wx_w16(rw + d, w); rw[d + 2u] = (uint8_t)(w >> 16);
                uint8_t* q = ml > 3u ? rw + d + 3u : dump; *q = (uint8_t)(w >> 24);
            }
        } else {
            if (ready && ml <= WX_COOP_LEN) {
                for (uint32_t k = 0; k < n1; k += 8) { const uint32_t m = n1 - k < 8u ? n1 - k : 8u; wx_wn(rw + d + k, wx_g64(out + src + k, m, src + k + 8u <= cap), m); }
                if (ml > n1) wx_lane_w2w(rw, d + n1, off, ml - n1);
            }
            for (unsigned long long lm = rm & longm; lm; lm &= lm - 1) {
                const int j = cz_unii(__ffsll((long long)lm) - 1);
                const uint32_t dj = cz_readlane(d, j), oj = cz_readlane(off, j), mj = cz_readlane(ml, j), nj = cz_readlane(n1, j);
                cz_wave_sync();
                if (nj) { wx_wave_g2w(rw, dj, out + (dj - oj), nj); cz_wave_sync(); }
                if (mj > nj) wx_wave_w2w(rw, dj + nj, oj, mj - nj);
            }
        }
        WX_FENCE();
        if (smallseq) { if (ready) wx_mark32(ctl.fin, opos - rbase, tot); }
        else {
            if (ready && tot <= 512u) wx_mark(ctl.fin, opos - rbase, tot);
            for (unsigned long long lm = rm & hugem; lm; lm &= lm - 1) {
                const int j = cz_unii(__ffsll((long long)lm) - 1);
                wx_wave_mark(ctl.fin, cz_readlane(opos, j) - rbase, cz_readlane(tot, j));
            }
        }
        if (ready) undone = 0;
        cz_wave_sync();
        if (first) break;
    }
    return __ballot(undone);
}

/* Chunks of one block's records (sequence_execution.cairo:12-83), all waves.  The entry of "chunk -1" holds the state before chunk 0.
 * A wave works on two chunks at a time, 16 apart: per turn it runs step 1 of the NEXT chunk y and publishes its sums, then moves
 * the data of the CURRENT chunk x (steps 3 and 4), then does the look-back of y (step 2) and issues y's literal loads.  So the
 * sums of a chunk are out a whole data phase before anybody's look-back asks for them, and literal bytes have a whole step 1 to
 * arrive. */
__device__ static void wx_block_sequences(WxCtl& ctl, uint8_t* ring, cz_gcptr out, cz_gcptr64 rec, uint32_t nseq, cz_gcptr bits, cz_gcptr lbase, uint32_t lit_len,
                                          int lit_rle, uint32_t rle_byte, uint32_t cap, uint32_t wave, uint32_t nwaves, uint32_t pstart, uint32_t c0, unsigned long long* wxp, const int poison) {
    /* The window holds the block's output from position `pstart` on: byte `pos` of the frame at rw[pos].  Chunks from c0 on (c0 > 0:
       the pass before this one filled the window; its bytes are in HBM now and the window starts again at the position before chunk
       c0).  The first chunk whose output would pass the end of the window sets ctl.reset_at; chunks from there on are left for
       the next pass. */
    uint8_t* const rw = ring - pstart;
    const uint32_t rbase = pstart, wlimit = cap;
#ifdef CZ_PROFILE
    unsigned long long wxt_ = 0;
#endif
    const uint32_t nch = (nseq + 63u) >> 6, lane = (uint32_t)LANE;
    const uint32_t yfirst = c0 + ((wave + nwaves - (c0 & (nwaves - 1u))) & (nwaves - 1u));   /* this wave's chunks: those equal to its number mod the waves */
    if (yfirst >= nch) return;
    auto load_rec = [&](uint32_t ch) -> uint64_t { const uint32_t i = 64u * ch + lane; return rec[i < nseq ? i : nseq - 1u]; };   /* coalesced 8-byte loads */
    uint64_t r1 = load_rec(yfirst), r2 = load_rec(yfirst + nwaves), r3 = load_rec(yfirst + 2u * nwaves);
    uint8_t* const dump = ctl.dump + 64u * wave + lane;
    /* step 1 results of chunk y, kept over the data phase of chunk x */
    uint32_t a_ll = 0, a_ml = 0, a_orel = 0, a_lrel = 0, a_asym = 0; int a_active = 0, a_bad = 0;
    uint32_t a_sum_ll = 0, a_sum_tot = 0, a_o0 = 0, a_o1 = 0, a_o2 = 0; int a_habs = 0;
    WxData x; x.ll = x.ml = x.opos = x.off = x.lpos = 0; x.lw = 0; x.active = 0;
    int x_ok = 0;                                                       /* chunk x passed its checks (else nothing of it is written: the frame is given up) */
    uint32_t out1 = pstart, out2 = pstart, out3 = pstart;               /* positions behind the chunk in hand (x), behind this wave's previous chunk (x - 16) and behind the one before (x - 32) */
    uint32_t y = yfirst; int have_x = 0;
    unsigned long long m_undone = 0;                                    /* matches of chunk x that are not written yet */
    for (;;) {
        const int have_y = y < nch && y < WX_UNI(wx_ld(&ctl.reset_at));
        if (have_y) {
            /* 1: values, sums, the chunk's history transform */
            const uint64_t r = r1;
            r1 = r2; r2 = r3; r3 = load_rec(y + 3u * nwaves);
            WX_PROF_T0(); WX_PROF_CNT(9);
            const WxStep1 s1 = wx_step1(ctl, r, bits, nseq, y);
            a_ll = s1.ll; a_ml = s1.ml; a_orel = s1.orel; a_lrel = s1.lrel; a_asym = s1.asym; a_active = s1.active; a_bad = s1.bad;
            a_sum_ll = s1.sum_ll; a_sum_tot = s1.sum_tot; a_o0 = s1.o0; a_o1 = s1.o1; a_o2 = s1.o2; a_habs = s1.habs;
            /* the chunk's sums — and its outgoing history when that does not depend on the incoming one (three pushes in 64
               sequences: nearly always) — are published at once */
            if (lane == 0 && !(poison && y == 2u)) {                    /* (CZ_DEBUG_WX_POISON: chunk 2 never publishes, every look-back behind it runs into its bound) */
                uint32_t* const mine = ctl.slot[y & (WX_NS - 1u)];
                if (a_habs) { wx_st(&mine[5], a_o0); wx_st(&mine[6], a_o1); wx_st(&mine[7], a_o2); }
                wx_st(&mine[1], a_sum_tot); wx_st(&mine[2], a_sum_ll);
                WX_FENCE();
                wx_st(&mine[0], ((y + 1u) << 3) | WX_F_AGG | (a_habs ? WX_F_HOK : 0u));
            }
            WX_PROF_ACC(0);
        }
        if (have_x && x_ok) {
            WX_PROF_T0();
            const uint32_t ll = x.ll, opos = x.opos;
            /* 3: literals: from the literal buffer (asked for at the end of the chunk's look-back) to their place in the window */
            {
                const unsigned long long l8 = __ballot(ll > 8u);
                if (__ballot(ll > 0u)) {
                    if (!__ballot(ll > 2u)) {
#pragma unroll
                        for (uint32_t j = 0; j < 2; j++) { uint8_t* q = j < ll ? rw + opos + j : dump; *q = (uint8_t)(x.lw >> (8 * j)); }
                    } else wx_wn(rw + opos, x.lw, ll < 8u ? ll : 8u);
                }
                if (l8) {
                    /* the rest of longer runs: by the lane up to WX_COOP_LEN, by the wave beyond */
                    const uint32_t lpos = x.lpos;
                    if (ll > 8u && ll <= WX_COOP_LEN) {
                        for (uint32_t k = 8; k < ll; k += 8) {
                            const uint32_t m = ll - k < 8u ? ll - k : 8u;
                            const uint64_t v = lit_rle ? 0x0101010101010101ull * rle_byte : wx_g64(lbase + lpos + k, m, lpos + k + 8u <= lit_len);
                            wx_wn(rw + opos + k, v, m);
                        }
                    }
                    for (unsigned long long lm = __ballot(ll > WX_COOP_LEN); lm; lm &= lm - 1) {
                        const int j = cz_unii(__ffsll((long long)lm) - 1);
                        const uint32_t n = cz_readlane(ll, j) - 8u, at = cz_readlane(opos, j) + 8u, from = cz_readlane(lpos, j) + 8u;
                        if (lit_rle) wx_wave_fill(rw, at, rle_byte, n); else wx_wave_g2w(rw, at, lbase + from, n);
                    }
                }
            }
            cz_wave_sync();
            WX_PROF_ACC(2);
            /* 4: matches: the first round here, what it leaves after the look-back of the next chunk (wx_match_rounds) */
            m_undone = wx_match_rounds(ctl, rw, out, rbase, cap, x, out3, dump, 1, wxp);
            WX_PROF_ACC(3);
        }
        if (!have_y) {
            if (m_undone) { WX_PROF_T0(); wx_match_rounds(ctl, rw, out, rbase, cap, x, out3, dump, 0, wxp, m_undone); WX_PROF_ACC(3); }
            break;
        }
        /* 2: the state before chunk y, by a decoupled look-back: the entries of the 16 chunks before it — the nearest one whose own
           state is known, plus the sums of those between.  The serial chain has one link per ROUND of chunks, not one per chunk; a
           history that is not absolute chains through its chunk only. */
        WX_PROF_T0();
        uint32_t P_in = 0, L_in = 0, h0 = 0, h1 = 0, h2 = 0; int stop = 0;
        {
            /* wait (one word per entry): the nearest chunk before y whose own state is known (chunk y - 16, this wave's previous
               one, at the latest), the sums of those between, and the history behind chunk y - 1 */
            const uint32_t* const ent = ctl.slot[(y - 1u - (lane & 15u)) & (WX_NS - 1u)];
            uint32_t k = 0, lbspins = 0;
            for (;;) {
                const uint32_t w0 = wx_ld(&ent[0]);
                const uint32_t f = lane < 16u && (w0 >> 3) == y - lane ? (w0 & 7u) : 0u;   /* (an entry may still be that of a chunk 64 earlier) */
                const unsigned long long inclm = __ballot(f & WX_F_INCL), aggm = __ballot(f & WX_F_AGG);
                if (inclm) {
                    k = (uint32_t)cz_unii(__ffsll((long long)inclm) - 1);
                    const unsigned long long below = (1ull << k) - 1ull;
                    if ((aggm & below) == below && (cz_readlane(f, 0) & WX_F_HOK)) break;
                }
                if (WX_UNI(wx_ld(&ctl.reset_at)) <= y) { stop = 1; break; }   /* a chunk before this one did not fit the window: the next pass */
                if (WX_UNI(wx_ld(&ctl.err))) { stop = 1; break; }            /* the frame has been given up */
                if (++lbspins > WX_SPIN_LIMIT) { if (lane == 0) wx_st(&ctl.err, 1u); stop = 1; break; }   /* bounded: give the frame up */
                WX_PAUSE(); WX_PROF_CNT(6);
                WX_SPIN_GUARD(lbspins, "WX SPIN look-back: y %u lane %u w0 %08x\n", y, lane, w0);
            }
            WX_FENCE();
            uint32_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (lane < 16u && !stop) wx_read_entry(ent, e);
            uint32_t sp = lane < k ? e[1] : (lane == k ? e[3] : 0u), sl = lane < k ? e[2] : (lane == k ? e[4] : 0u);
#define WX_ROW_ADD(CTRL) do { sp += cz_dpp<CTRL, 0xF>(0u, sp); sl += cz_dpp<CTRL, 0xF>(0u, sl); } while (0)
            WX_ROW_ADD(CZ_DPP_SHR1); WX_ROW_ADD(CZ_DPP_SHR2); WX_ROW_ADD(CZ_DPP_SHR4); WX_ROW_ADD(CZ_DPP_SHR8);
#undef WX_ROW_ADD
            P_in = cz_readlane(sp, 15); L_in = cz_readlane(sl, 15);
            h0 = cz_readlane(e[5], 0); h1 = cz_readlane(e[6], 0); h2 = cz_readlane(e[7], 0);
        }
        /* (a position that has left the buffer stays where it is: no wrap-around can bring it back inside) */
        const uint32_t P_out = P_in > wlimit ? P_in : P_in + a_sum_tot, L_out = L_in > lit_len ? L_in : L_in + a_sum_ll;
        if (!stop && P_out <= wlimit && P_out - rbase > WX_RING) {
            /* the chunk does not fit the window any more: the next pass starts with it — or, when it would not fit an empty window
               either (64 sequences of more than 128 KiB: matches of tens of KB), wx_slow_chunk does it between two passes */
            if (lane == 0) { if (P_out - P_in > WX_RING) WX_MIN(&ctl.slow_at, y); WX_MIN(&ctl.reset_at, y); }
            stop = 1;
        }
        if (stop) {
            if (m_undone) { wx_match_rounds(ctl, rw, out, rbase, cap, x, out3, dump, 0, wxp, m_undone); m_undone = 0; }
            break;
        }
        if (lane == 0 && !(poison && y == 2u)) {
            uint32_t* const mine = ctl.slot[y & (WX_NS - 1u)];
            if (!a_habs) { wx_st(&mine[5], wx_resolve(a_o0, h0, h1, h2)); wx_st(&mine[6], wx_resolve(a_o1, h0, h1, h2)); wx_st(&mine[7], wx_resolve(a_o2, h0, h1, h2)); }
            wx_st(&mine[4], L_out); wx_st(&mine[3], P_out);
            WX_FENCE();
            wx_st(&mine[0], ((y + 1u) << 3) | WX_F_AGG | WX_F_HOK | WX_F_INCL);
        }
        /* every check of execute_sequences (sequence_execution.cairo:28-36 literals, :47 zero offset; decode_buffer.cairo:65 offset
           beyond the output so far — no dictionary here) and the capacity of the caller's buffer */
        WxData nx;
        nx.ll = a_ll; nx.ml = a_ml; nx.active = a_active;
        nx.off = wx_resolve(a_asym, h0, h1, h2);
        nx.opos = P_in + a_orel; nx.lpos = L_in + a_lrel;
        const int bad = a_bad | (a_active && (nx.off - 1u >= nx.opos + a_ll));
        const int nx_ok = !(__ballot(bad) || P_out > wlimit || P_out - rbase > WX_RING || L_out > lit_len);
        if (!nx_ok && lane == 0) { wx_st(&ctl.err, 1u); WX_DBG("WX give-up: chunk %u P_in %u P_out %u wlimit %u L_out %u lit_len %u badmask %llx\n", y, P_in, P_out, wlimit, L_out, lit_len, (unsigned long long)__builtin_popcountll(0)); }
        nx.lw = 0x0101010101010101ull * rle_byte;
        if (nx_ok && !lit_rle && a_ll > 0u) nx.lw = wx_g64(lbase + nx.lpos, a_ll, nx.lpos + 8u <= lit_len);
        WX_PROF_ACC(1);
        /* the matches of chunk x that had to wait: the chunks they wait for have had this look-back's time */
        if (m_undone) { wx_match_rounds(ctl, rw, out, rbase, cap, x, out3, dump, 0, wxp, m_undone); m_undone = 0; WX_PROF_ACC(3); }
        x = nx; x_ok = nx_ok;
        out3 = out2; out2 = out1; out1 = P_out;
        have_x = 1; y += nwaves;
    }
}

/* A chunk whose output is longer than the window (ctl.slow_at), by the whole workgroup, between two passes: the window has gone to
 * HBM up to the position before the chunk; wave 0 does steps 1 and 2 (the state before the chunk is in the entry of chunk cs - 1,
 * complete since the barrier behind the pass) and leaves the 64 sequences in `tab` (the empty window's first bytes); then the
 * sequences go one after the other, literals and match copied by all threads straight in the frame's output (a match that
 * overlaps itself by doubling: decode_buffer.cairo:95-127 is periodic from its source on); the entry of the chunk gets the state
 * behind it, and the next pass starts with an empty window there.  Returns 0 when a check of execute_sequences fails (the frame
 * is given up).  Every thread of the workgroup calls it. */
__device__ static int wx_slow_chunk(WxCtl& ctl, uint32_t* tab, cz_gptr out, cz_gcptr64 rec, uint32_t nseq, cz_gcptr bits, cz_gcptr lbase, uint32_t lit_len,
                                    int lit_rle, uint32_t rle_byte, uint32_t cap, uint32_t cs, uint32_t tid, uint32_t nthreads) {
    const uint32_t lane = (uint32_t)LANE;
    const uint32_t cnt = nseq - 64u * cs < 64u ? nseq - 64u * cs : 64u;
    WX_AT(1);
    if (tid < 64u) {
        const uint32_t i = 64u * cs + lane;
        const WxStep1 s1 = wx_step1(ctl, rec[i < nseq ? i : nseq - 1u], bits, nseq, cs);
        const uint32_t* const e = ctl.slot[(cs - 1u) & (WX_NS - 1u)];
        const uint32_t P_in = cz_uni(wx_ld(&e[3])), L_in = cz_uni(wx_ld(&e[4])), h0 = cz_uni(wx_ld(&e[5])), h1 = cz_uni(wx_ld(&e[6])), h2 = cz_uni(wx_ld(&e[7]));
        const uint32_t off = wx_resolve(s1.asym, h0, h1, h2), opos = P_in + s1.orel, lpos = L_in + s1.lrel;
        const int bad = s1.bad | (s1.active && (off - 1u >= opos + s1.ll));   /* sequence_execution.cairo:47, decode_buffer.cairo:65 */
        const uint64_t P_out = (uint64_t)P_in + s1.sum_tot, L_out = (uint64_t)L_in + s1.sum_ll;
        const int ok = !(__ballot(bad) || P_out > cap || L_out > lit_len);   /* :28-36, and the caller's buffer */
        tab[lane] = s1.ll; tab[64u + lane] = s1.ml; tab[128u + lane] = off; tab[192u + lane] = opos; tab[256u + lane] = lpos;
        if (lane == 0) {
            ctl.slow_ok = (uint32_t)ok; ctl.slow_P = (uint32_t)P_out; ctl.slow_L = (uint32_t)L_out;
            uint32_t* const mine = ctl.slot[cs & (WX_NS - 1u)];
            wx_st(&mine[5], wx_resolve(s1.o0, h0, h1, h2)); wx_st(&mine[6], wx_resolve(s1.o1, h0, h1, h2)); wx_st(&mine[7], wx_resolve(s1.o2, h0, h1, h2));
            wx_st(&mine[1], s1.sum_tot); wx_st(&mine[2], s1.sum_ll); wx_st(&mine[4], (uint32_t)L_out); wx_st(&mine[3], (uint32_t)P_out);
        }
    }
    __syncthreads();
    if (tid == 0) WX_DBG("WX slow chunk %u: ok %u, position behind it %u\n", cs, ctl.slow_ok, ctl.slow_P);
    if (!cz_uni(ctl.slow_ok)) return 0;
    for (uint32_t q = 0; q < cnt; q++) {
        WX_AT(100 + q);
        const uint32_t ll = cz_uni(tab[q]), ml = cz_uni(tab[64u + q]), off = cz_uni(tab[128u + q]), opos = cz_uni(tab[192u + q]), lpos = cz_uni(tab[256u + q]);
        for (uint32_t i = 16u * tid; i < ll; i += 16u * nthreads) {    /* literals: literal buffer -> output */
            const uint32_t m = ll - i < 16u ? ll - i : 16u;
            uint4 v;
            if (lit_rle) { const uint32_t w = 0x01010101u * rle_byte; v = uint4{w, w, w, w}; }
            else v = cz_load_upto16(lbase + lpos + i, m, lpos + i + 16u <= lit_len);
            (cz_store_upto16)(out + opos + i, v, m);
        }
        if (!ml) continue;
        __syncthreads();                                                /* (the source may be this sequence's literals) */
        const uint32_t d = opos + ll;
        uint32_t copied = 0, dist = off;
        while (copied < ml) {
            WX_AT(200 + q);
            while (dist < 16u * nthreads && 2u * dist <= off + copied) dist += dist;
            const uint32_t step = ml - copied < dist ? ml - copied : dist;
            for (uint32_t i = 16u * tid; i < step; i += 16u * nthreads) {   /* sources [.., + step) lie below the destinations: a 16-byte load stays inside them */
                const uint32_t m = step - i < 16u ? step - i : 16u;
                const uint4 v = cz_load_upto16((cz_gcptr)out + (d + copied - dist + i), m, m == 16u);
                (cz_store_upto16)(out + d + copied + i, v, m);
            }
            __syncthreads();                                            /* the waves of a workgroup share their CU's L1: the barrier orders these stores before the next loads */
            copied += step;
        }
    }
    __syncthreads();
    WX_AT(3);
    if (tid == 0) wx_st(&ctl.slot[cs & (WX_NS - 1u)][0], ((cs + 1u) << 3) | WX_F_AGG | WX_F_HOK | WX_F_INCL);
    return 1;
}

/* The early launch, at a large block: waits until cz_chain_kernel's launch of the large blocks has published the block (header word 3:
   1 done, 2 given up), then makes this CU see what that kernel wrote (the caller's workgroup barrier lets the other waves go on).
   The wait is bounded like every other wait of this kernel: past it the frame goes to cz_decode_frames_kernel.  One thread calls it. */
__device__ static inline int wx_wait_chain(const cz_batch_args& a, uint64_t hdr) {
    uint32_t polls = 0; uint64_t v;
    while ((v = CZ_LD_AGENT(&a.chain_arena[hdr + 3])) == 0) {
        if (++polls > WX_CHAIN_WAIT_POLLS) return 0;
#ifdef CZ_EMU
        sched_yield();
#else
        __builtin_amdgcn_s_sleep(64);
#endif
    }
    CZ_ACQUIRE_AGENT();
    return v == 1;
}

/* One workgroup per frame; frames from the list cz_scan_kernel made (a.wx_list, scan_ctl[206] entries). */
extern "C" __global__ void __launch_bounds__(WX_THREADS, 1) cz_wexec_kernel(cz_batch_args a) {
    uint8_t* const ring = (uint8_t*)cz_dyn_lds;
    WxCtl& ctl = *(WxCtl*)(ring + WX_RING);
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, nwaves = nthreads >> 6, wave = cz_uni(tid >> 6);
    const uint32_t nlist = cz_uni(a.scan_ctl[206]);
    const int big_only = cz_wx_big_only(a);                             /* near-offset batch: only its large frames (CZ_PRE_WXBIG), and all of those */
    /* The EARLY launch (args.early == 1) starts behind the small blocks' chains and the literal kernels, while the launch of the large
       blocks' chains still runs: it takes the batch's large frames (a batch arranged that way only) and executes each block by
       block, waiting for a large block's chain at the block (wx_wait_chain).  What it decides from — the sums cz_chain_kernel
       takes from the code tables — may still be growing: whatever it and the later launch decide, a frame is executed by whoever
       claims it. */
    const int early = a.early == 1u;
    if (nlist == 0 || !((!early && cz_wx_side_by_side(a)) || big_only)) return;
#ifndef CZ_EXP_NO_WXCOUNT
    {   /* the launch has a workgroup for every CU and more; args.wx_cus of them stay (half of the chip: profiles/r4 wexec_sweep) */
        /* (the slot goes through a word of WxCtl: a static __shared__ variable would shift the dynamic LDS — the window — off its alignment: 0.12 ms on config 4a) */
        if (tid == 0) ctl.fidx = atomicAdd(&a.scan_ctl[early ? 214 : 213], 1u);
        __syncthreads();
        const uint32_t wx_slot = cz_uni(ctl.fidx);
        __syncthreads();
        if (a.wx_cus && wx_slot >= a.wx_cus) return;
    }
#endif
    unsigned long long wxp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef CZ_PROFILE
    unsigned long long wxt_ = 0;
#endif
    for (uint32_t i = tid; i < 36; i += nthreads) ctl.llml[i] = CZ_LL_BASE[i] | ((uint32_t)CZ_LL_BITS[i] << 24);
    for (uint32_t i = tid; i < 53; i += nthreads) ctl.llml[40 + i] = CZ_ML_BASE[i] | ((uint32_t)CZ_ML_BITS[i] << 24);
    CzsWalk w; CzsBlk b;                                                /* thread 0's walk over the frame's headers (the scan's own walker: same decisions) */
    czs_begin(w, nullptr, 0, 0);
    b.type = b.blk_off = b.bsize = b.lt = b.regen = b.lit_hdr = b.nseq = b.sbody = b.modes = 0;
    uint64_t chain_cursor = 0, lit_cursor = 0; uint32_t pre_blocks = 0;
    for (;;) {
        __syncthreads();
        WX_PROF_T0();
        if (tid == 0) {
            uint32_t f;
            for (;;) {
                const uint32_t i = atomicAdd(a.wx_counter, 1u);
                f = i < nlist ? a.wx_list[i] : 0xFFFFFFFFu;
                if (f == 0xFFFFFFFFu || !big_only || (a.frame_pre[f] & CZ_PRE_WXBIG)) break;
            }
            ctl.fidx = f; ctl.go = 2; ctl.err = 0;
            if (f != 0xFFFFFFFFu) {
                /* still as the scan left it?  (cz_chain_kernel / the huff0 kernels may have taken the frame back) */
                const uint32_t pre = atomicOr(&a.frame_pre[f], CZ_PRE_CLAIMED);   /* (cz_execute_frames_kernel runs beside this kernel: whoever claims a frame first does it) */
                chain_cursor = a.frame_first[f]; lit_cursor = a.lit_first[f]; pre_blocks = pre & CZ_PRE_COUNT;
                if (!(pre & CZ_PRE_CLAIMED)) atomicAdd(&a.scan_ctl[208], 1u);
                if (pre & (CZ_PRE_CLAIMED | CZ_PRE_LISTED)) ctl.go = 3;   /* the other kernel's, or handed back already (a huff0 kernel took it back and listed it) */
                else if ((pre & CZ_PRE_REGULAR) && !(pre & CZ_PRE_DONE) && chain_cursor != 0 && lit_cursor != 0 && a.out_cap[f] < 0x80000000ull) {
                    czs_begin(w, a.in_base + a.in_off[f], a.in_len[f], 1);
                    if (w.active) { ctl.go = 1; ctl.P = 0; ctl.blocks = 0; ctl.hist[0] = 1; ctl.hist[1] = 4; ctl.hist[2] = 8; }   /* scratch.cairo:35 */
                }
            }
        }
        __syncthreads();
        const uint32_t f = cz_uni(ctl.fidx);
        if (f == 0xFFFFFFFFu) break;
        if (cz_uni(ctl.go) != 1u) {                                     /* taken by the other kernel (3), or not what the scan left (2): cz_decode_frames_kernel does it from scratch */
            if (tid == 0 && ctl.go == 2u && a.fallback_list) { atomicAdd(&a.scan_ctl[207], 1u); cz_list_fallback(a, f); }
            continue;
        }
        cz_gcptr fsrc = (cz_gcptr)(a.in_base + a.in_off[f]);
        cz_gptr out = (cz_gptr)(a.out_base + a.out_off[f]);
        const uint32_t cap = (uint32_t)a.out_cap[f];
        int ok = 0;
        for (;;) {
            /* thread 0: the next block (block_decoder.cairo:237-321 and the section headers behind it) */
            if (tid == 0) {
                uint32_t go = 2;
                const int has = czs_next(w, a.chain_min_nseq, b);
                if (!has) go = w.ok ? 0u : 2u;
                else {
                    const uint32_t P = ctl.P, blocks = ctl.blocks;
                    const int predone = blocks < pre_blocks;
                    go = 1; ctl.btype = b.type; ctl.bsize = b.bsize; ctl.blt = b.lt; ctl.bregen = b.regen; ctl.bnseq = b.nseq; ctl.bpredone = (uint32_t)predone;
                    const uint64_t sp = (uint64_t)(uintptr_t)(fsrc + b.blk_off);
                    ctl.src_lo = (uint32_t)sp; ctl.src_hi = (uint32_t)(sp >> 32);
                    uint32_t outlen = b.type != 2 ? b.bsize : (b.nseq ? 0u : b.regen);
                    if (P + outlen > cap || outlen > WX_RING) go = 2;      /* (a block's output must fit the window; the format's blocks do: block_decoder.cairo:67) */
                    if (go == 1 && b.type == 2 && !predone) {
                        /* the block's literals (literals_section_decoder.cairo:32-56): Raw in place, RLE one byte, Huffman-coded in the node the scan laid out */
                        uint64_t lp = sp + b.lit_hdr; uint32_t rle = 0, byte = 0;
                        if (b.lt == 1) { rle = 1; byte = fsrc[b.blk_off + b.lit_hdr]; }
                        else if (b.lt >= 2) {
                            if (lit_cursor < 64) go = 2;
                            else {
                                CZ_GLOBAL const uint64_t* node = (CZ_GLOBAL const uint64_t*)(a.lit_arena + lit_cursor);
                                const uint64_t nxt = node[0], meta = node[1];
                                if ((uint32_t)meta != b.regen) go = 2;
                                lp = (uint64_t)(uintptr_t)(a.lit_arena + lit_cursor + 16); lit_cursor = nxt;
                            }
                        }
                        ctl.lit_lo = (uint32_t)lp; ctl.lit_hi = (uint32_t)(lp >> 32); ctl.lit_rle = rle; ctl.lit_byte = byte; ctl.lit_len = b.regen;
                        if (go == 1 && b.nseq && chain_cursor && early && cz_hbs(b.nseq) >= CZ_BIG_BLOCK_CLASS && !wx_wait_chain(a, chain_cursor)) go = 2;
                        if (go == 1 && b.nseq) {
                            if (!chain_cursor) go = 2;
                            else {
                                /* cz_chain_kernel's header: {nseq << 32 | maps that follow, offset of the bitstream, next header}, maps, records */
                                cz_gcptr64 arena = (cz_gcptr64)a.chain_arena;
                                const uint64_t w0 = arena[chain_cursor], w1 = arena[chain_cursor + 1], w2 = arena[chain_cursor + 2];
                                const uint64_t mp = (uint64_t)(uintptr_t)(a.chain_arena + chain_cursor + 4), rp = mp + 8ull * CZ_CHAIN_MAP_WORDS;
                                const uint64_t bp = sp + (uint32_t)w1;
                                chain_cursor = w2;
                                if ((uint32_t)(w0 >> 32) != b.nseq) go = 2;
                                ctl.mapflags = (uint32_t)w0; ctl.maps_lo = (uint32_t)mp; ctl.maps_hi = (uint32_t)(mp >> 32);
                                ctl.rec_lo = (uint32_t)rp; ctl.rec_hi = (uint32_t)(rp >> 32); ctl.bits_lo = (uint32_t)bp; ctl.bits_hi = (uint32_t)(bp >> 32);
                                /* the state before chunk 0: the entry of "chunk -1" */
                                for (uint32_t k = 0; k < WX_NS; k++) ctl.slot[k][0] = 0;
                                uint32_t* s0 = ctl.slot[WX_NS - 1u];
                                s0[1] = 0; s0[2] = 0; s0[3] = P; s0[4] = 0; s0[5] = ctl.hist[0]; s0[6] = ctl.hist[1]; s0[7] = ctl.hist[2];
                                s0[0] = WX_F_AGG | WX_F_HOK | WX_F_INCL;
                            }
                        }
                    }
                }
                ctl.go = go;
                if (go == 2) WX_DBG("WX give-up at block %u: type %u size %u nseq %u P %u has %d ok %d chain %llu lit %llu\n", ctl.blocks, b.type, b.bsize, b.nseq, ctl.P, has, w.ok, (unsigned long long)chain_cursor, (unsigned long long)lit_cursor);
            }
            __syncthreads();
            WX_PROF_ACC(4);
            const uint32_t go = cz_uni(ctl.go);
            if (go != 1u) { ok = go == 0u; break; }
            const uint32_t btype = cz_uni(ctl.btype), bsize = cz_uni(ctl.bsize), nseq = cz_uni(ctl.bnseq), regen = cz_uni(ctl.bregen);
            const uint32_t P = cz_uni(ctl.P);
            const int predone = (int)cz_uni(ctl.bpredone);
            cz_gcptr bsrc = (cz_gcptr)(uintptr_t)wx_ptr(cz_uni(ctl.src_lo), cz_uni(ctl.src_hi));
            uint32_t P_end = P, wbase = P;                              /* the window holds this block's output from wbase on: byte `pos` of the frame at (ring - wbase)[pos] */
            uint8_t* const rw = ring - P;
            if (predone) P_end = P + (btype != 2 ? bsize : regen);      /* cz_tile_kernel / cz_huf_kernel put this block's output in place */
            else if (btype == 0) { wx_wg_g2w(rw, P, bsrc, bsize, tid, nthreads); P_end = P + bsize; }            /* Raw, block_decoder.cairo:97-103 */
            else if (btype == 1) { wx_wg_fill(rw, P, bsrc[0], bsize, tid, nthreads); P_end = P + bsize; }        /* RLE :104-123 */
            else {
                cz_gcptr lbase = (cz_gcptr)(uintptr_t)wx_ptr(cz_uni(ctl.lit_lo), cz_uni(ctl.lit_hi));
                const int lit_rle = (int)cz_uni(ctl.lit_rle); const uint32_t rle_byte = cz_uni(ctl.lit_byte), lit_len = cz_uni(ctl.lit_len);
                uint32_t L_end = 0;
                if (nseq) {
                    const uint32_t mapflags = cz_uni(ctl.mapflags);
                    cz_gcptr mp = (cz_gcptr)(uintptr_t)wx_ptr(cz_uni(ctl.maps_lo), cz_uni(ctl.maps_hi));
                    if (tid < 80u) {                                    /* the maps of the tables this block defined: 80 x 16 bytes */
                        const uint32_t which = tid < 32u ? 1u : (tid < 64u ? 4u : 2u);
                        if (mapflags & which) { const uint4 v = cz_ldu128(mp + 16u * tid); *(uint4*)(ctl.maps + 16u * tid) = v; }
                    }
                    __syncthreads();
                    const cz_gcptr64 recp = (cz_gcptr64)(uintptr_t)wx_ptr(cz_uni(ctl.rec_lo), cz_uni(ctl.rec_hi));
                    const cz_gcptr bitp = (cz_gcptr)(uintptr_t)wx_ptr(cz_uni(ctl.bits_lo), cz_uni(ctl.bits_hi));
                    const uint32_t nch = (nseq + 63u) >> 6;
                    uint32_t c0 = 0; int give_up = 0;
                    for (;;) {
                        /* one pass = as many chunks as fit the window; nothing of the pass is final yet */
                        for (uint32_t i = tid; i < WX_RING / 32u + 4u; i += nthreads) ctl.fin[i] = 0;
                        if (tid == 0) { ctl.reset_at = WX_INF; ctl.slow_at = WX_INF; }
                        __syncthreads();
                        wx_block_sequences(ctl, ring, (cz_gcptr)out, recp, nseq, bitp, lbase, lit_len, lit_rle, rle_byte, cap, wave, nwaves, wbase, c0, wxp, (a.debug_flags & CZ_DEBUG_WX_POISON) != 0u);
                        __syncthreads();
                        WX_PROF_T0();
                        WX_AT(4);
                        const uint32_t cs = cz_uni(ctl.reset_at);
                        const int slow = cs != WX_INF && cz_uni(ctl.slow_at) == cs;
                        if (cz_uni(wx_ld(&ctl.err)) || (cs != WX_INF && ((cs <= c0 && !slow) || cs >= nch))) { give_up = 1; break; }
                        if (cs == WX_INF) break;
                        /* the window is full: its bytes go to HBM, and it starts again at the position before chunk cs (whose entries are
                           cleared of what the pass published of them: a chunk's sums say its wave has got there) */
                        const uint32_t pmid = cz_uni(ctl.slot[(cs - 1u) & (WX_NS - 1u)][3]);
                        wx_wg_flush(ring - wbase, wbase, out + wbase, pmid - wbase, tid, nthreads);
                        if (tid < WX_NS && (ctl.slot[tid][0] >> 3) > cs) ctl.slot[tid][0] = 0;
                        wbase = pmid; c0 = cs;
                        WX_AT(5);
                        __syncthreads();
                        if (slow) {                                     /* chunk cs is longer than the window: straight in the frame's output */
                            if (!wx_slow_chunk(ctl, (uint32_t*)ring, out, recp, nseq, bitp, lbase, lit_len, lit_rle, rle_byte, cap, cs, tid, nthreads)) { give_up = 1; break; }
                            __syncthreads();
                            wbase = cz_uni(ctl.slow_P); c0 = cs + 1u;
                            if (c0 >= nch) break;
                        }
                    }
                    if (give_up) { ok = 0; break; }
                    const uint32_t* se = ctl.slot[(nch - 1u) & (WX_NS - 1u)];   /* the last chunk's entry */
                    P_end = cz_uni(se[3]); L_end = cz_uni(se[4]);
                    if (P_end > cap || L_end > lit_len) { ok = 0; break; }
                    if (tid == 0) { ctl.hist[0] = se[5]; ctl.hist[1] = se[6]; ctl.hist[2] = se[7]; }
                }
                /* literals behind the last sequence — for a block without sequences, all of them (sequence_execution.cairo:72-78, block_decoder.cairo:229-232) */
                const uint32_t rest = lit_len - L_end;
                if (P_end + rest > cap || rest > WX_RING) { ok = 0; break; }
                if (P_end + rest - wbase > WX_RING) {                   /* no room behind the sequences' output: that goes to HBM first */
                    wx_wg_flush(ring - wbase, wbase, out + wbase, P_end - wbase, tid, nthreads);
                    wbase = P_end;
                    __syncthreads();
                }
                if (rest) { if (lit_rle) wx_wg_fill(ring - wbase, P_end, rle_byte, rest, tid, nthreads); else wx_wg_g2w(ring - wbase, P_end, lbase + L_end, rest, tid, nthreads); }
                P_end += rest;
            }
            __syncthreads();
            if (!predone) wx_wg_flush(ring - wbase, wbase, out + wbase, P_end - wbase, tid, nthreads);
            /* (later blocks read these bytes from HBM: the barrier at the top of the loop orders the stores before those loads — the
               waves of a workgroup share their CU's L1, which writes through) */
            if (tid == 0) { ctl.P = P_end; ctl.blocks += 1; }
            WX_PROF_ACC(5);
        }
        __syncthreads();
        if (ok && tid == 0) {
            /* frame_decoder.cairo:189-200: the last block was the frame's last; the stored content checksum follows it */
            uint32_t ck = 0, fl = CZ_RESULT_FINISHED; uint64_t pos = w.end; int good = 1;
            if (w.has_checksum) { if (w.len - w.end >= 4) { ck = (uint32_t)czs_ld8(w.src, w.len, pos); fl |= CZ_RESULT_HAS_CHECKSUM; pos += 4; } else good = 0; }
            if (good) {
                cz_frame_result r; r.status = 0; r.blocks_decoded = ctl.blocks; r.bytes_consumed = pos; r.bytes_produced = ctl.P; r.checksum_from_data = ck; r.flags = fl;
                r.detail[0] = ctl.blocks; r.detail[1] = pos; r.calculated_checksum = 0; r.reserved = 0;
                a.results[f] = r;
                a.frame_pre[f] |= CZ_PRE_WXDONE;
                atomicAdd(&a.scan_ctl[209], 1u);
            } else ok = 0;
        }
        if (!ok && tid == 0 && a.fallback_list) { atomicAdd(&a.scan_ctl[207], 1u); cz_list_fallback(a, f); }   /* cz_decode_frames_kernel does it from scratch and reports what is wrong with it */
    }
#ifdef CZ_PROFILE
    if ((tid & 63u) == 0 && a.prof) for (int i = 0; i < 10; i++) atomicAdd(&a.prof[40 + i], wxp[i]);
#endif
}
