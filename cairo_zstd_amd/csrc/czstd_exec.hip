/*
 * czstd_exec.hip — cz_exec_frames_kernel: execution of frames whose FSE chains cz_chain_kernel already ran.
 *
 * cz_decode_frames_kernel gives a frame one wave and keeps the frame's window in HBM: on blocks
 * of many short matches (BASELINE config 4a: 32 768 three-byte matches at uniformly random offsets)
 * every match pulls a memory sector from one of thousands of live windows.  Here a frame gets a whole
 * compute unit — one workgroup of CZX_WAVES (16) waves — and the block's output is assembled in a
 * 128 KiB LDS ring (160 KB of LDS per CU hold a whole 128 KiB block) and written to HBM once,
 * coalesced.  A block's sequences are cut into chunks of 64 (one per lane) and the chunks dealt to the
 * waves round robin; per supergroup of up to 512 chunks (whose output fits the ring):
 *   pass 1  every wave, for each of its chunks: codes -> (ll, ml, offset_value) from the chain record,
 *           the chunk's literal / output byte counts and its repeat-offset transform in SYMBOLIC form
 *           (each history slot afterwards = an entering slot minus a constant, or a constant;
 *           sequence_execution.cairo:85-129) -> chunk summary in LDS
 *   pass 2  one wave scans the summaries (8 per lane, then a wave scan whose operator composes
 *           transforms): every chunk learns its literal position, output position and entering history
 *   pass 3  every wave, for each of its chunks: resolves offsets with the entering history, checks what
 *           execute_sequences / DecodeBuffer::repeat check (sequence_execution.cairo:12-66,
 *           decode_buffer.cairo:62-133), copies literals (global -> LDS) and matches (LDS -> LDS, or
 *           global -> LDS for sources that left the ring).  Chunks RETIRE in order: a watermark in LDS
 *           says up to which output byte everything is written; a match whose source lies above it waits.
 *   flush   the supergroup's bytes go from the ring to HBM with 16-byte stores.
 * Huffman literals are decoded by four of the waves (one stream each, 64 bit ranges per stream, the
 * self-synchronising scheme of cz_huf_streams_par) while the other waves already run pass 1.
 *
 * Like the chain pre-pass this kernel is a pure accelerator: it only finishes frames that decode
 * without any error; on ANY irregularity (malformed section, literal-stream error, execution error,
 * output too small, a chunk whose output exceeds the ring, ...) it leaves the frame untouched for
 * cz_decode_frames_kernel, which decodes it from scratch and reports the reference's status code.  A frame
 * finished here is marked frame_first[f] = CZX_DONE and skipped there.
 */
#ifndef CZX_WAVES
#define CZX_WAVES 16
#endif
#define CZX_THREADS (64 * CZX_WAVES)
#define CZX_WIN_BYTES 131072u
#define CZX_WIN_MASK (CZX_WIN_BYTES - 1u)
#define CZX_MAX_CHUNKS 512u
#define CZX_DONE 0xFFFFFFFFFFFFFFFFull
#define WAVE ((int)(threadIdx.x >> 6))

/* chunk summary; pass 2 turns it into the chunk's entering state in place */
struct CzxSum {
    uint32_t a;      /* pass 1: literal bytes of the chunk        pass 2: literal position of the chunk (in the block's literals) */
    uint32_t b;      /* pass 1: output bytes of the chunk         pass 2: output position of the chunk (frame position) */
    uint32_t sel;    /* pass 1: transform selectors, 2 bits/slot  pass 2: output bytes of the chunk */
    uint32_t v[3];   /* pass 1: transform values                  pass 2: entering history */
};
/* dynamic LDS of this kernel: [maps 5120 B | window 128 KiB | summaries 12 KiB | control] */
#define CZX_LDS_WIN_OFF CZ_FSE_LDS_BYTES
#define CZX_LDS_SUM_OFF (CZX_LDS_WIN_OFF + CZX_WIN_BYTES)
#define CZX_LDS_CTL_OFF (CZX_LDS_SUM_OFF + CZX_MAX_CHUNKS * 24u)
#define CZX_LDS_BYTES (CZX_LDS_CTL_OFF + 64u)
#define CZX_WIN ((uint8_t*)cz_dyn_lds + CZX_LDS_WIN_OFF)
#define CZX_SUMS ((CzxSum*)((uint8_t*)cz_dyn_lds + CZX_LDS_SUM_OFF))
#define CZX_CTL ((volatile uint32_t*)((uint8_t*)cz_dyn_lds + CZX_LDS_CTL_OFF))
enum { CZX_C_WM = 0, CZX_C_PUNT = 1, CZX_C_NSG = 2, CZX_C_SGEND = 3, CZX_C_LITEND = 4, CZX_C_FRAME = 5, CZX_C_WMC = 6 };

/* Diagnostic build only (-DCZ_PROFILE): thread 0 accumulates s_memtime deltas per phase of this kernel in args.prof[40..]:
 * 40 headers + section parse, 41 Huffman table, 42 Huffman streams, 43 maps, 44 pass 1, 45 pass 2, 46 pass 3, 47 flush, 48 raw/rle/tail copies */
#ifdef CZ_PROFILE
#define CZX_PROF_DECL unsigned long long czx_t_ = __builtin_amdgcn_s_memtime(); unsigned long long czx_p_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CZX_PROF(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); czx_p_[i] += n_ - czx_t_; czx_t_ = n_; } while (0)
#define CZX_PROF_FLUSH() do { if (threadIdx.x == 0 && a.prof) for (int i_ = 0; i_ < 10; i_++) atomicAdd(&a.prof[40 + i_], czx_p_[i_]); } while (0)
#else
#define CZX_PROF_DECL
#define CZX_PROF(i) do { } while (0)
#define CZX_PROF_FLUSH() do { } while (0)
#endif
__device__ static inline void czx_punt() { CZX_CTL[CZX_C_PUNT] = 1u; }
__device__ static inline void czx_sleep() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_sleep(1);
#elif defined(CZ_EMU)
    sched_yield();                              /* tests/emu: lanes are preemptive threads */
#endif
}

/* symbolic repeat-offset transform: slot k afterwards = (sel_k < 3 ? entering slot sel_k - val_k : val_k) */
struct CzxXf { uint32_t sel, v0, v1, v2; };
__device__ static inline uint32_t czx_xf_val(const CzxXf& f, uint32_t k) { return k == 0 ? f.v0 : (k == 1 ? f.v1 : f.v2); }
/* first f, then g */
__device__ static inline CzxXf czx_xf_compose(const CzxXf& f, const CzxXf& g) {
    CzxXf r; uint32_t sel = 0, out[3];
#pragma unroll
    for (uint32_t k = 0; k < 3; k++) {
        const uint32_t gs = (g.sel >> (2 * k)) & 3u, gv = czx_xf_val(g, k);
        uint32_t s, v;
        if (gs == 3) { s = 3; v = gv; }
        else {
            const uint32_t fs = (f.sel >> (2 * gs)) & 3u, fv = czx_xf_val(f, gs);
            s = fs; v = fs == 3 ? fv - gv : fv + gv;
        }
        sel |= s << (2 * k); out[k] = v;
    }
    r.sel = sel; r.v0 = out[0]; r.v1 = out[1]; r.v2 = out[2];
    return r;
}
#define CZX_XF_ID 0x24u   /* slot k = entering slot k */

/* codes -> values for the record of this lane (sequence_section_decoder.cairo:239-256) */
__device__ static inline void czx_decode_record(uint64_t r, int active, cz_gcptr bits, uint32_t& ll, uint32_t& ml, uint32_t& ov) {
    const uint8_t* mapll = (const uint8_t*)CZ_FSE_LL; const uint8_t* mapml = mapll + 512; const uint8_t* mapof = mapll + 1024;
    ll = 0; ml = 0; ov = 4;
    if (active) {
        const uint32_t xt = (uint32_t)r, st = (uint32_t)(r >> 32);
        const uint32_t oc = mapof[(st >> 18) & 255];
        const uint32_t tl = sh.b.c.llml[mapll[st & 511]], tm = sh.b.c.llml[40 + mapml[(st >> 9) & 511]];
        const uint32_t mx = tm >> 24, lx = tl >> 24;
        if (!(st & CZC_REC_WIDE)) {
            ov = (1u << oc) + __builtin_amdgcn_ubfe(xt, 32 - oc, oc);   /* :243 */
            ml = (tm & 0xFFFFFFu) + __builtin_amdgcn_ubfe(xt, 32 - oc - mx, mx);         /* :249-256 */
            ll = (tl & 0xFFFFFFu) + __builtin_amdgcn_ubfe(xt, 32 - oc - mx - lx, lx);
        } else {                                                        /* more than 32 extra bits: the record says where they are in the bitstream */
            const uint64_t W = cz_stream_window64(bits, xt);
            ov = (1u << oc) + cz_field(W, 0, oc);
            ml = (tm & 0xFFFFFFu) + cz_field(W, oc, mx);
            ll = (tl & 0xFFFFFFu) + cz_field(W, oc + mx, lx);
        }
    }
}

/* The wave scan of cz_history, shared by the symbolic (pass 1) and the concrete (pass 3) resolution:
 * T, V = transform of lanes 0..lane (see cz_history for the encoding). */
__device__ static inline void czx_history_scan(uint32_t cnt, uint32_t ll, uint32_t ov, uint32_t& T, uint32_t& V) {
    const int active = (uint32_t)LANE < cnt;
    const uint32_t kind = !active ? 0u : (ov > 3 ? 3u : ov - (ll > 0 ? 1u : 0u));
    const uint32_t t01 = (kind & 1u) ? 0x00020001u : CZ_T_ID, t23 = (kind & 1u) ? 0x00010004u : 0x00010002u;
    T = (kind & 2u) ? t23 : t01;
    V = (uint32_t)LANE;
#define CZX_HT_STEP(CTRL, RM) do { const uint32_t pT = cz_dpp<CTRL, RM>(CZ_T_ID, T), pV = cz_dpp<CTRL, RM>(0u, V); \
        const uint32_t R = __builtin_amdgcn_perm(T, pT, T); V = __builtin_amdgcn_perm(V, pV, T); \
        const uint32_t m = ((R >> 2) & 0x00010101u) * 0xFFu; T = (0x00060504u & m) | (R & ~m); } while (0)
    CZX_HT_STEP(CZ_DPP_SHR1, 0xF); CZX_HT_STEP(CZ_DPP_SHR2, 0xF); CZX_HT_STEP(CZ_DPP_SHR4, 0xF); CZX_HT_STEP(CZ_DPP_SHR8, 0xF);
    CZX_HT_STEP(CZ_DPP_BCAST15, 0xA); CZX_HT_STEP(CZ_DPP_BCAST31, 0xC);
#undef CZX_HT_STEP
}
/* pass 1: the chunk's transform in symbolic form (uniform result) */
__device__ static inline CzxXf czx_history_symbolic(uint32_t cnt, uint32_t ll, uint32_t ov) {
    uint32_t T, V;
    czx_history_scan(cnt, ll, ov, T, V);
    const int active = (uint32_t)LANE < cnt;
    const int dec = active && ov == 3 && ll == 0;
    /* value pushed by this lane: a constant, or for the h0 - 1 sequences (slot 0 before the lane) - 1 */
    uint32_t ps = 3, pv = ov - 3;
    const unsigned long long dm = __ballot(dec);
    if (dm) {
        const uint32_t eT = cz_dpp<CZ_DPP_WAVE_SHR1, 0xF>(CZ_T_ID, T), eV = cz_dpp<CZ_DPP_WAVE_SHR1, 0xF>(0u, V);
        for (unsigned long long m = dm; m; m &= m - 1) {
            const int j = cz_unii(__ffsll((long long)m) - 1);
            const uint32_t bT = cz_readlane(eT, j) & 0xFFu, bV = cz_readlane(eV, j) & 63u;
            const uint32_t qs = cz_readlane(ps, cz_unii((int)bV)), qv = cz_readlane(pv, cz_unii((int)bV));
            const uint32_t s = (bT & 4u) ? qs : (bT & 3u), v = (bT & 4u) ? qv : 0u;
            if (LANE == j) { ps = s; pv = s == 3 ? v - 1 : v + 1; }
        }
    }
    const int lastl = cz_unii((int)cnt - 1);
    const uint32_t fT = cz_readlane(T, lastl), fV = cz_readlane(V, lastl);
    CzxXf f; uint32_t sel = 0, out[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t tk = (fT >> (8 * k)) & 0xFFu, vk = (fV >> (8 * k)) & 63u;
        const uint32_t qs = cz_readlane(ps, cz_unii((int)vk)), qv = cz_readlane(pv, cz_unii((int)vk));
        const uint32_t s = (tk & 4u) ? qs : (tk & 3u), v = (tk & 4u) ? qv : 0u;
        sel |= s << (2 * k); out[k] = v;
    }
    f.sel = cz_uni(sel); f.v0 = cz_uni(out[0]); f.v1 = cz_uni(out[1]); f.v2 = cz_uni(out[2]);
    return f;
}

/* one byte of the frame's output at frame position p: the ring holds positions >= res_lo */
__device__ static inline uint32_t czx_out_byte(cz_gcptr out, uint32_t p, uint32_t res_lo) {
    return p >= res_lo ? CZX_WIN[p & CZX_WIN_MASK] : (uint32_t)__builtin_nontemporal_load(out + p);
}

/* Huffman literal streams of a block by the waves 0..3 (one stream per wave; a single stream: wave 0),
 * 64 bit ranges per stream.  Same scheme and same contract as cz_huf_streams_par; every lane of the
 * participating waves calls it.  Results in sh.bc.st_count / st_flags of the wave's stream. */
__device__ static void czx_huf_stream_wave(cz_gcptr blk, cz_gptr target, uint32_t k, uint32_t cap) {
    CzBroadcast& bc = sh.bc;
    const uint32_t mb = cz_uni(sh.huf_max_bits);
    const uint32_t i = (uint32_t)LANE;
    const uint8_t* S = blk + bc.stream_off[k]; const uint32_t len = bc.stream_len[k]; const uint8_t* E = S + len;
    const uint32_t lastb = len ? E[-1] : 0;
    const int padbad = lastb == 0;
    const int32_t P0 = padbad ? 0 : (int32_t)len * 8 - (int32_t)(__clz((int)lastb) - 24 + 1);
    uint32_t m = (uint32_t)P0 >> 8; m = m < 1 ? 1 : (m > 64 ? 64 : m);   /* ranges of >= 256 bits, at most 64 */
    const int32_t C = (P0 + (int32_t)m - 1) / (int32_t)m;
    const int live = !padbad && i < m && P0 > 0;
    const int32_t top_b = P0 - (int32_t)i * C;
    const int32_t stop = (i + 1 == m) ? 0 : P0 - (int32_t)(i + 1) * C;
    CzGBits g; g.S = (uintptr_t)S; g.E = (uintptr_t)E; g.LB = (uintptr_t)blk; g.p = top_b;
    int32_t s = top_b, e = stop; uint32_t n = 0;
    n = cz_gb_decode(g, mb, stop, live, nullptr, 0, 0);           /* 1. speculative pass */
    if (live) e = g.p;
    for (int round = 0; round < 65; round++) {                          /* 2. fix the starts until nothing moves */
        const int32_t pe = __shfl_up(e, 1u);
        const int changed = live && i > 0 && pe != s;
        if (!__ballot(changed)) break;
        if (changed) { s = pe; g.p = s; }
        const uint32_t nred = cz_gb_decode(g, mb, stop, changed, nullptr, 0, 0);
        if (changed) { n = nred; e = g.p; }
    }
    const uint32_t incl = cz_wave_incl_scan(live ? n : 0u);             /* 3. output offsets, writing pass */
    const uint32_t total = cz_readlane(incl, 63), off = incl - (live ? n : 0u);
    const int32_t e_last = (int32_t)cz_readlane((uint32_t)e, cz_unii((int)m - 1));
    {
        g.p = s;
        const uint32_t room = off < cap ? cap - off : 0;
        cz_gb_decode(g, mb, stop, live, target + off, room, 0);
    }
    if (i == 0) {
        uint32_t fl = padbad ? 1u : 0u;
        if (!padbad && e_last != 0) fl |= 2u;
        const uint32_t cnt = padbad ? 0 : total;
        bc.st_count[k] = cnt; bc.st_flags[k] = fl | ((cnt != cap) ? 4u : 0u);
    }
}

/* workgroup-wide copy / fill, global -> global: every wave takes a 4 KiB-aligned share */
__device__ static void czx_wg_copy(cz_gptr dst, cz_gcptr src, uint32_t n) {
    const uint32_t per = ((n + CZX_WAVES - 1) / CZX_WAVES + 4095u) & ~4095u;
    const uint32_t lo = (uint32_t)WAVE * per;
    if (lo < n) cz_coop_copy((uint8_t*)dst + lo, (const uint8_t*)src + lo, n - lo < per ? n - lo : per);
}
__device__ static void czx_wg_fill(cz_gptr dst, uint8_t byte, uint32_t n) {
    const uint32_t per = ((n + CZX_WAVES - 1) / CZX_WAVES + 4095u) & ~4095u;
    const uint32_t lo = (uint32_t)WAVE * per;
    if (lo < n) cz_coop_fill((uint8_t*)dst + lo, byte, n - lo < per ? n - lo : per);
}

/* ring -> HBM for frame positions [lo, hi) */
__device__ static void czx_flush(cz_gptr out, uint32_t lo, uint32_t hi) {
    const uint32_t head_end = ((lo + 15u) & ~15u) < hi ? ((lo + 15u) & ~15u) : hi;
    for (uint32_t p = lo + threadIdx.x; p < head_end; p += CZX_THREADS) out[p] = CZX_WIN[p & CZX_WIN_MASK];
    const uint32_t body_end = head_end + ((hi - head_end) & ~15u);
    for (uint32_t p = head_end + 16u * threadIdx.x; p < body_end; p += 16u * CZX_THREADS) {
        const uint4 v = *(const uint4*)(CZX_WIN + (p & CZX_WIN_MASK));
        __builtin_memcpy((uint8_t*)out + p, &v, 16);
    }
    for (uint32_t p = body_end + threadIdx.x; p < hi; p += CZX_THREADS) out[p] = CZX_WIN[p & CZX_WIN_MASK];
}

struct CzxBlock {            /* uniform per block */
    cz_gptr out; uint32_t cap;
    uint32_t blk_start;      /* frame position where the block's output starts */
    CzLit lit;
};

/* LDS atomics on the control words / chunk flags */
__device__ static inline void czx_atomic_max(volatile uint32_t* p, uint32_t v) {
#if defined(CZ_EMU)
    uint32_t cur = __atomic_load_n((uint32_t*)p, __ATOMIC_SEQ_CST);
    while (cur < v && !__atomic_compare_exchange_n((uint32_t*)p, &cur, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { }
#else
    atomicMax((uint32_t*)p, v);
#endif
}
#define CZX_DONE_BIT 0x80000000u
__device__ static inline uint32_t czx_chunk_done(uint32_t j) { return ((volatile CzxSum*)CZX_SUMS)[j].sel >> 31; }

/* pass 3 for one chunk (all lanes of one wave).  c = index of the chunk in its supergroup of nsg chunks, base = frame
 * position of the chunk, lbase = its literal position, res_lo = lowest frame position the ring holds during this
 * supergroup, ctot = output bytes of the chunk, sg_start / sg_end = frame positions of the supergroup.
 * Chunks finish in any order: a chunk sets its DONE flag when all its bytes are in the ring; CTL[WM] = position up to
 * which every chunk is done (a cheap "certainly written" test); a match whose source lies above it looks up the
 * chunks that produce its source (start positions of the 64 previous chunks, one per lane) and waits for exactly those.
 * Returns nonzero on any failed check (1) or when another wave gave the frame up (2). */
__device__ static int czx_exec_chunk(const CzxBlock& B, uint32_t c, uint32_t nsg, uint32_t cnt, uint32_t ll, uint32_t ml, uint32_t off,
                                     uint32_t base, uint32_t lbase, uint32_t res_lo, uint32_t ctot, uint32_t sg_start, uint32_t sg_end) {
    const int active = (uint32_t)LANE < cnt;
    if (!active) { ll = 0; ml = 0; off = 1; }
    const uint32_t incl_ll = cz_wave_incl_scan(ll), tot = ll + ml, incl_tot = cz_wave_incl_scan(tot);
    const uint32_t orel = incl_tot - tot, lrel = incl_ll - ll;
    const uint32_t lpos = lbase + lrel, opos = base + orel, dst = opos + ll;
    /* sequence_execution.cairo:28-36, :47; decode_buffer.cairo:65; the output capacity */
    const int bad = active && ((ll > 0 && lpos + ll > B.lit.len) || off == 0 || (ml > 0 && off > dst) || dst + ml > B.cap);
    if (__ballot(bad)) return 1;
    uint8_t* win = CZX_WIN;
    /* literals */
    {
        const unsigned long long longm = __ballot(ll > 8);
        if (ll > 0 && ll <= 8) {
            if (B.lit.rle) for (uint32_t k = 0; k < ll; k++) win[(opos + k) & CZX_WIN_MASK] = B.lit.byte;
            else {
                cz_gcptr s = B.lit.p + lpos;
                uint8_t t[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) if (k < ll) t[k] = __builtin_nontemporal_load(s + k);
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) if (k < ll) win[(opos + k) & CZX_WIN_MASK] = t[k];
            }
        }
        for (unsigned long long m = longm; m; m &= m - 1) {             /* long runs: the whole wave copies */
            const int j = cz_unii(__ffsll((long long)m) - 1);
            const uint32_t n = cz_readlane(ll, j), o = cz_readlane(opos, j), lp = cz_readlane(lpos, j);
            for (uint32_t k = (uint32_t)LANE; k < n; k += 64) win[(o + k) & CZX_WIN_MASK] = B.lit.rle ? B.lit.byte : __builtin_nontemporal_load(B.lit.p + lp + k);
        }
    }
    /* matches.  Source [src, src + span), span = min(off, ml) (a longer match repeats that with period off). */
    const uint32_t src = dst - off, span = off < ml ? off : ml, send = src + span;   /* send <= dst */
    const uint32_t ext = send < base ? send : base;                     /* end of the part of the source that earlier chunks produce */
    int done = !(active && ml > 0);
    /* sources that left the ring: HBM (flushed before this supergroup began) */
    if (!done && send <= res_lo) {
        uint32_t idx = 0;
        for (uint32_t k = 0; k < ml; k++) { win[(dst + k) & CZX_WIN_MASK] = __builtin_nontemporal_load(B.out + src + idx); idx = idx + 1 == off ? 0 : idx + 1; }
        done = 1;
    }
    /* which chunks produce the part of the source that is neither certainly written nor inside this chunk */
    uint32_t jcur = 0, jend = 0;                                        /* chunks [jcur, jend) must be done */
    {
        const uint32_t wm0 = cz_readlane(CZX_CTL[CZX_C_WM], 0);
        const unsigned long long need = __ballot(!done && src < base && ext > wm0);
        if (need) {
            const uint32_t pstart = (uint32_t)LANE < c ? ((volatile CzxSum*)CZX_SUMS)[c - 1 - (uint32_t)LANE].b : sg_start;   /* lane l: start of chunk c - 1 - l */
            for (unsigned long long m = need; m; m &= m - 1) {
                const int p = cz_unii(__ffsll((long long)m) - 1);
                const uint32_t qlo = cz_readlane(src, p), qhi = cz_readlane(ext, p) - 1u;
                const uint32_t dlo = (uint32_t)__popcll(__ballot((uint32_t)LANE < c && pstart > qlo)), dhi = (uint32_t)__popcll(__ballot((uint32_t)LANE < c && pstart > qhi));
                /* chunk c - 1 - d starts at or below q (d == min(c, 64): q is older than the chunks the lanes hold: wait for the watermark) */
                if (LANE == p) { jcur = dlo >= c || dlo >= 64 ? 0xFFFFFFFFu : c - 1 - dlo; jend = c - dhi; }
            }
        }
    }
    cz_wave_sync();
    for (;;) {
        const unsigned long long pend = __ballot(!done);
        if (!pend) break;
        const uint32_t wm = cz_readlane(CZX_CTL[CZX_C_WM], 0);
        if (__ballot(CZX_CTL[CZX_C_PUNT] != 0)) return 2;
        int extok = ext <= wm || src >= base;
        if (!done && !extok && jcur != 0xFFFFFFFFu) {
            while (jcur < jend && czx_chunk_done(jcur)) jcur++;
            extok = jcur >= jend;
        }
        const int f = cz_unii(__ffsll((long long)pend) - 1);
        const uint32_t Wl = cz_readlane(dst, f);                        /* this chunk: all earlier lanes are done, and all literals are in */
        const int ready = !done && extok && send <= Wl;
        if (!__ballot(ready)) continue;
        if (ready && ml <= 32) {
            if (ml <= 8 && off >= ml && src >= res_lo) {                /* the common short case: all loads, then all stores */
                uint8_t t[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) if (k < ml) t[k] = win[(src + k) & CZX_WIN_MASK];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) if (k < ml) win[(dst + k) & CZX_WIN_MASK] = t[k];
            } else {
                uint32_t idx = 0;
                for (uint32_t k = 0; k < ml; k++) { win[(dst + k) & CZX_WIN_MASK] = (uint8_t)czx_out_byte(B.out, src + idx, res_lo); idx = idx + 1 == off ? 0 : idx + 1; }
            }
            done = 1;
        }
        for (unsigned long long m = __ballot(ready && ml > 32); m; m &= m - 1) {   /* long matches: the whole wave copies */
            const int j = cz_unii(__ffsll((long long)m) - 1);
            const uint32_t n = cz_readlane(ml, j), o = cz_readlane(off, j), d = cz_readlane(dst, j);
            if (o >= 64) {                                              /* generations of 64 bytes never read what they write */
                for (uint32_t b0 = 0; b0 < n; b0 += 64) {
                    const uint32_t k = b0 + (uint32_t)LANE;
                    uint32_t v = 0;
                    if (k < n) v = czx_out_byte(B.out, d - o + k, res_lo);
                    if (k < n) win[(d + k) & CZX_WIN_MASK] = (uint8_t)v;
                    cz_wave_sync();
                }
            } else for (uint32_t k = (uint32_t)LANE; k < n; k += 64) win[(d + k) & CZX_WIN_MASK] = (uint8_t)czx_out_byte(B.out, d - o + k % o, res_lo);
            if (LANE == j) done = 1;
        }
        cz_wave_sync();
    }
    /* this chunk is done; move the watermark over every chunk that is */
    cz_wave_sync();
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0xC07F);                                 /* lgkmcnt(0): this wave's ring writes have landed */
#endif
    if (LANE == 0) {
        ((volatile CzxSum*)CZX_SUMS)[c].sel = ctot | CZX_DONE_BIT;
        uint32_t w = CZX_CTL[CZX_C_WMC];                                /* the flag is set first, then looked at: of two chunks finishing together at least one sees the other */
        const uint32_t w0 = w;
        while (w < nsg && czx_chunk_done(w)) w++;
        if (w != w0) {
            czx_atomic_max(&CZX_CTL[CZX_C_WMC], w);
            czx_atomic_max(&CZX_CTL[CZX_C_WM], w < nsg ? ((volatile CzxSum*)CZX_SUMS)[w].b : sg_end);
        }
    }
    return 0;
}

/* one frame; every thread of the workgroup.  Returns 0 when the frame is finished (results written). */
__device__ static int czx_run_frame(const cz_batch_args& a, uint32_t f, cz_gptr lit_scratch) {
    CzBroadcast& bc = sh.bc;
    cz_gcptr src = (cz_gcptr)(a.in_base + a.in_off[f]); const uint64_t src_len = a.in_len[f];
    cz_gptr out = (cz_gptr)(a.out_base + a.out_off[f]);
    const uint64_t cap64 = a.out_cap[f];
    if (cap64 >= 0x7FFF0000ull || src_len >= 0x7FFF0000ull) return 1;
    const uint32_t cap = (uint32_t)cap64;
    __syncthreads();
    if (threadIdx.x == 0) { bc.d0 = 0; bc.d1 = 0; bc.err = cz_parse_frame_header(src, src_len, bc); CZX_CTL[CZX_C_PUNT] = 0; }
    __syncthreads();
    if (cz_unii(bc.err)) return 1;
    uint32_t pos = cz_uni(bc.hdr_len); const uint32_t has_checksum = cz_uni(bc.has_checksum);
    __syncthreads();
    uint32_t produced = 0, blocks = 0, flags = 0, cksum = 0;
    CZX_PROF_DECL;
    uint64_t cursor = cz_uni64(a.frame_first[f]);
    cz_gcptr64 arena = (cz_gcptr64)a.chain_arena;
    cz_state_reset();
    for (;;) {
        if ((uint32_t)src_len - pos < 3) return 1;
        const uint32_t b0 = src[pos], b1 = src[pos + 1], b2 = src[pos + 2];
        const uint32_t btype = (b0 >> 1) & 3, bsize = (b0 >> 3) | (b1 << 5) | (b2 << 13), blast = b0 & 1;
        if (btype == 3 || bsize > 128u * 1024u) return 1;
        const uint32_t body = pos + 3, content = btype == 1 ? 1u : bsize;
        if ((uint32_t)src_len - body < content) return 1;
        cz_gcptr blk = src + body;
        if (btype == 0) {
            if (produced + bsize > cap) return 1;
            czx_wg_copy(out + produced, blk, bsize);
            produced += bsize;
        } else if (btype == 1) {
            if (produced + bsize > cap) return 1;
            czx_wg_fill(out + produced, blk[0], bsize);
            produced += bsize;
        } else {
            /* ---- compressed block (block_decoder.cairo:139-235) */
            const uint32_t stage_hi = bsize < 512 ? bsize : 512;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < stage_hi; i += CZX_THREADS) sh.a.t1.stage[i] = blk[i];
            __syncthreads();
            if (threadIdx.x == 0) bc.err = cz_parse_sections(blk, bsize, stage_hi);
            __syncthreads();
            if (cz_unii(bc.err) == CZ_PARSE_NEED_WTAB) {
                if (WAVE == 0) cz_huf_weight_table();
                __syncthreads();
                if (threadIdx.x == 0) bc.err = cz_parse_sections(blk, bsize, stage_hi, 0, 1);
                __syncthreads();
            }
            if (cz_unii(bc.err)) return 1;
            const uint32_t huf_fill = cz_uni(bc.huf_fill), lt = cz_uni(bc.lit_type), regen = cz_uni(bc.regen), lit_total = cz_uni(bc.lit_total);
            const uint32_t nseq = cz_uni(bc.nseq), nstreams = cz_uni(bc.nstreams), huf_nsym = cz_uni(bc.huf_nsym);
            if (cz_unii(bc.seq_hdr_err)) return 1;
            __syncthreads();
            CZX_PROF(0);
            if (huf_fill) {                                             /* huff0_decoder.cairo:451-463, a symbol per wave at a time */
                if (WAVE == 0) cz_huf_rank_wave(huf_nsym);
                __syncthreads();
                const uint32_t max_bits = sh.huf_max_bits;
                for (uint32_t s = (uint32_t)WAVE; s < huf_nsym; s += CZX_WAVES) {
                    const uint32_t b = sh.b.c.hbits[s];
                    if (!b) continue;
                    const uint32_t hb = sh.b.c.sym_base[s], len = 1u << (max_bits - b);
                    const uint16_t e = (uint16_t)(s | (b << 8));
                    for (uint32_t k = (uint32_t)LANE; k < len; k += 64) sh.a.huf[hb + k] = e;
                }
            } else if (lt == 3) return 1;                               /* Treeless: left to cz_decode_frames_kernel (it carries the table) */
            __syncthreads();
            CZX_PROF(1);
            CzxBlock B; B.out = out; B.cap = cap; B.blk_start = produced;
            B.lit.rle = 0; B.lit.byte = 0; B.lit.len = regen; B.lit.p = blk;
            /* chain records of this block */
            cz_gcptr64 rec = nullptr; uint32_t mapflags = 0; cz_gcptr seqbits = blk;
            if (nseq) {
                if (!cursor) return 1;
                const uint64_t w0 = arena[cursor], w1 = arena[cursor + 1], w2 = arena[cursor + 2];
                seqbits = blk + cz_uni((uint32_t)w1);
                cz_gcptr64 maps = arena + cursor + 4;
                rec = maps + CZ_CHAIN_MAP_WORDS;
                cursor = cz_uni64(w2);
                mapflags = cz_uni((uint32_t)w0);
                if (cz_uni((uint32_t)(w0 >> 32)) != nseq) return 1;
                uint8_t* mapll = (uint8_t*)CZ_FSE_LL;
                for (uint32_t i = threadIdx.x; i < 80; i += CZX_THREADS) {          /* 1280 bytes in 16-byte pieces: LL 0..31, ML 32..63, OF 64..79 */
                    const uint32_t t = i < 32 ? 0u : (i < 64 ? 2u : 1u);
                    if ((mapflags >> t) & 1u) { uint4 v; __builtin_memcpy(&v, (cz_gcptr)maps + 16u * i, 16); *(uint4*)(mapll + 16u * i) = v; }
                }
            }
            CZX_PROF(3);
            /* literals (literals_section_decoder.cairo:32-56): Huffman streams on waves 0..3 */
            if (lt == 0) B.lit.p = blk + (lit_total - regen);
            else if (lt == 1) { B.lit.rle = 1; B.lit.byte = blk[lit_total - 1]; }
            else {
                if (regen > CZ_LIT_SCRATCH_BYTES) return 1;
                cz_gptr target = lit_scratch;
                if (nseq == 0) { if (produced + regen > cap) return 1; target = out + produced; }
                if (nstreams == 4) {
                    const uint32_t seg = (regen + 3) >> 2;
                    if (3 * seg > regen) return 1;
                    if (WAVE < 4) czx_huf_stream_wave(blk, target + (uint32_t)WAVE * seg, (uint32_t)WAVE, WAVE < 3 ? seg : regen - 3 * seg);
                } else if (WAVE == 0) czx_huf_stream_wave(blk, target, 0, regen);
                B.lit.p = target;
            }
            __syncthreads();
            if (lt >= 2) {
                const uint32_t ns = nstreams == 4 ? 4u : 1u;
                uint32_t tot = 0, fl = 0;
                for (uint32_t k = 0; k < ns; k++) { tot += bc.st_count[k]; fl |= bc.st_flags[k]; }
                if (nstreams != 4) fl &= ~6u;                           /* a single stream has no end test (:118-170) and its count is checked below */
                if (cz_uni(fl) || cz_uni(tot) != regen) return 1;
                /* the literal bytes were written by other waves of this CU: make them visible to every wave's loads */
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_s_waitcnt(0x0F70);                     /* vmcnt(0) */
#endif
                __syncthreads();
            }
            CZX_PROF(2);
            if (nseq == 0) {                                            /* block_decoder.cairo:229-232 */
                if (lt < 2) {
                    if (produced + regen > cap) return 1;
                    if (B.lit.rle) czx_wg_fill(out + produced, B.lit.byte, regen); else czx_wg_copy(out + produced, B.lit.p, regen);
                }
                produced += regen;
            } else {
                /* ---- sequences: supergroups of up to CZX_MAX_CHUNKS chunks */
                const uint32_t nchunks = (nseq + 63) >> 6;
                uint32_t c0 = 0, lit_used = 0;
                uint32_t h0 = cz_uni(sh.hist[0]), h1 = cz_uni(sh.hist[1]), h2 = cz_uni(sh.hist[2]);
                while (c0 < nchunks) {
                    const uint32_t nc = nchunks - c0 < CZX_MAX_CHUNKS ? nchunks - c0 : CZX_MAX_CHUNKS;
                    /* pass 1 */
                    for (uint32_t c = (uint32_t)WAVE; c < nc; c += CZX_WAVES) {
                        const uint32_t first = (c0 + c) << 6, cnt = nseq - first < 64 ? nseq - first : 64;
                        const int active = (uint32_t)LANE < cnt;
                        const uint64_t r = active ? rec[first + (uint32_t)LANE] : 0;
                        uint32_t ll, ml, ov;
                        czx_decode_record(r, active, seqbits, ll, ml, ov);
                        const CzxXf xf = czx_history_symbolic(cnt, ll, ov);
                        const uint32_t sl = cz_readlane(cz_wave_incl_scan(ll), 63), st = cz_readlane(cz_wave_incl_scan(ll + ml), 63);
                        if (LANE == 0) { CzxSum& s = CZX_SUMS[c]; s.a = sl; s.b = st; s.sel = xf.sel; s.v[0] = xf.v0; s.v[1] = xf.v1; s.v[2] = xf.v2; }
                    }
                    __syncthreads();
                    CZX_PROF(4);
                    /* pass 2 (wave 0): 8 summaries per lane, wave scan over (sums, composed transforms) */
                    if (WAVE == 0) {
                        const uint32_t lo = (uint32_t)LANE * 8u;
                        CzxXf acc; acc.sel = CZX_XF_ID; acc.v0 = acc.v1 = acc.v2 = 0;
                        uint32_t sa = 0, sb = 0;
                        for (uint32_t k = 0; k < 8; k++) if (lo + k < nc) {
                            const CzxSum s = CZX_SUMS[lo + k];
                            CzxXf g; g.sel = s.sel; g.v0 = s.v[0]; g.v1 = s.v[1]; g.v2 = s.v[2];
                            acc = czx_xf_compose(acc, g); sa += s.a; sb += s.b;
                        }
                        /* exclusive scan over lanes */
                        CzxXf run = acc; uint32_t ra = sa, rb = sb;
                        for (int d = 1; d < 64; d <<= 1) {
                            CzxXf p; p.sel = __shfl_up(run.sel, (unsigned)d); p.v0 = __shfl_up(run.v0, (unsigned)d); p.v1 = __shfl_up(run.v1, (unsigned)d); p.v2 = __shfl_up(run.v2, (unsigned)d);
                            const uint32_t pa = __shfl_up(ra, (unsigned)d), pb = __shfl_up(rb, (unsigned)d);
                            if (LANE >= d) { run = czx_xf_compose(p, run); ra += pa; rb += pb; }
                        }
                        CzxXf ex; ex.sel = __shfl_up(run.sel, 1u); ex.v0 = __shfl_up(run.v0, 1u); ex.v1 = __shfl_up(run.v1, 1u); ex.v2 = __shfl_up(run.v2, 1u);
                        uint32_t ea = __shfl_up(ra, 1u), eb = __shfl_up(rb, 1u);
                        if (LANE == 0) { ex.sel = CZX_XF_ID; ex.v0 = ex.v1 = ex.v2 = 0; ea = 0; eb = 0; }
                        /* how many chunks fit the ring: output positions only grow, so the chunks that fit are a prefix */
                        uint32_t nfit = 0;
                        {
                            uint32_t pb = eb;
                            for (uint32_t k = 0; k < 8; k++) if (lo + k < nc) {
                                const uint32_t cb = CZX_SUMS[lo + k].b;
                                if (pb + cb <= CZX_WIN_BYTES) nfit++;
                                pb += cb;
                            }
                        }
                        const uint32_t nsg = cz_readlane(cz_wave_incl_scan(nfit), 63);
                        /* entering state of every chunk */
                        {
                            CzxXf cur = ex; uint32_t pa = ea, pb = eb;
                            for (uint32_t k = 0; k < 8; k++) if (lo + k < nc) {
                                CzxSum& s = CZX_SUMS[lo + k];
                                CzxXf g; g.sel = s.sel; g.v0 = s.v[0]; g.v1 = s.v[1]; g.v2 = s.v[2];
                                const uint32_t ca = s.a, cb = s.b;
                                uint32_t e[3];
#pragma unroll
                                for (uint32_t q = 0; q < 3; q++) { const uint32_t ss = (cur.sel >> (2 * q)) & 3u, vv = czx_xf_val(cur, q); e[q] = ss == 3 ? vv : cz_pick3(ss, h0, h1, h2) - vv; }
                                s.a = lit_used + pa; s.b = produced + pb; s.sel = cb; s.v[0] = e[0]; s.v[1] = e[1]; s.v[2] = e[2];
                                cur = czx_xf_compose(cur, g); pa += ca; pb += cb;
                                if (lo + k + 1 == nsg) {                /* state after the supergroup */
                                    uint32_t o[3];
#pragma unroll
                                    for (uint32_t q = 0; q < 3; q++) { const uint32_t ss = (cur.sel >> (2 * q)) & 3u, vv = czx_xf_val(cur, q); o[q] = ss == 3 ? vv : cz_pick3(ss, h0, h1, h2) - vv; }
                                    sh.hist[0] = o[0]; sh.hist[1] = o[1]; sh.hist[2] = o[2];
                                    CZX_CTL[CZX_C_SGEND] = produced + pb; CZX_CTL[CZX_C_LITEND] = lit_used + pa;
                                }
                            }
                        }
                        if (LANE == 0) { CZX_CTL[CZX_C_NSG] = nsg; CZX_CTL[CZX_C_WM] = produced; CZX_CTL[CZX_C_WMC] = 0; }
                    }
                    __syncthreads();
                    CZX_PROF(5);
                    const uint32_t nsg = cz_uni(CZX_CTL[CZX_C_NSG]);
                    if (nsg == 0) return 1;                             /* a chunk larger than the ring */
                    const uint32_t sg_end = cz_uni(CZX_CTL[CZX_C_SGEND]), lit_end = cz_uni(CZX_CTL[CZX_C_LITEND]);
                    if (sg_end > cap) return 1;
                    const uint32_t res_lo = sg_end > CZX_WIN_BYTES ? (sg_end - CZX_WIN_BYTES > B.blk_start ? sg_end - CZX_WIN_BYTES : B.blk_start) : B.blk_start;
                    /* pass 3 */
                    int err = 0;
                    for (uint32_t c = (uint32_t)WAVE; c < nsg && !err; c += CZX_WAVES) {
                        const uint32_t first = (c0 + c) << 6, cnt = nseq - first < 64 ? nseq - first : 64;
                        const int active = (uint32_t)LANE < cnt;
                        const uint64_t r = active ? rec[first + (uint32_t)LANE] : 0;
                        uint32_t ll, ml, ov;
                        czx_decode_record(r, active, seqbits, ll, ml, ov);
                        const CzxSum s = CZX_SUMS[c];
                        uint32_t e0 = cz_uni(s.v[0]), e1 = cz_uni(s.v[1]), e2 = cz_uni(s.v[2]);
                        const uint32_t actual = cz_history(cnt, ll, ov, e0, e1, e2);
                        err = czx_exec_chunk(B, c, nsg, cnt, ll, ml, actual, cz_uni(s.b), cz_uni(s.a), res_lo, cz_uni(s.sel) & 0x7FFFFFFFu, produced, sg_end);
                        if (err == 1) czx_punt();
                    }
                    __syncthreads();
                    CZX_PROF(6);
                    if (cz_uni(CZX_CTL[CZX_C_PUNT])) return 1;
                    czx_flush(out, produced, sg_end);
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_s_waitcnt(0x0F70);                 /* vmcnt(0): flushed bytes may be read back as far sources */
#endif
                    __syncthreads();
                    h0 = cz_uni(sh.hist[0]); h1 = cz_uni(sh.hist[1]); h2 = cz_uni(sh.hist[2]);
                    produced = sg_end; lit_used = lit_end; c0 += nsg;
                    __syncthreads();
                    CZX_PROF(7);
                }
                /* remaining literals (sequence_execution.cairo:72-78) */
                if (lit_used < regen) {
                    const uint32_t rest = regen - lit_used;
                    if (produced + rest > cap) return 1;
                    if (B.lit.rle) czx_wg_fill(out + produced, B.lit.byte, rest); else czx_wg_copy(out + produced, B.lit.p + lit_used, rest);
                    produced += rest;
                }
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_s_waitcnt(0x0F70);                             /* vmcnt(0): this block's bytes are the next block's window */
#endif
        __syncthreads();
        CZX_PROF(8);
        pos = body + content; blocks++;
        if (blast) {
            flags |= CZ_RESULT_FINISHED;
            if (has_checksum) {
                if ((uint32_t)src_len - pos < 4) return 1;
                cksum = (uint32_t)src[pos] | ((uint32_t)src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16) | ((uint32_t)src[pos + 3] << 24);
                flags |= CZ_RESULT_HAS_CHECKSUM; pos += 4;
            }
            break;
        }
    }
    if (threadIdx.x == 0) {
        CZ_GLOBAL cz_frame_result* res = (CZ_GLOBAL cz_frame_result*)&a.results[f];
        res->status = 0; res->blocks_decoded = blocks; res->bytes_consumed = pos; res->bytes_produced = produced;
        res->checksum_from_data = cksum; res->flags = flags; res->calculated_checksum = 0; res->reserved = 0;
        res->detail[0] = blocks; res->detail[1] = pos;
    }
    CZX_PROF_FLUSH();
    return 0;
}

/* Persistent grid, one workgroup per CU: every workgroup pulls frames off a counter and takes those that have chain records. */
extern "C" __global__ void __launch_bounds__(CZX_THREADS, 1) cz_exec_frames_kernel(cz_batch_args a) {
    cz_init_llml();
    cz_gptr lit_scratch = (cz_gptr)(a.lit_scratch + (uint64_t)blockIdx.x * a.lit_scratch_stride);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) CZX_CTL[CZX_C_FRAME] = atomicAdd(a.exec_counter, 1u);
        __syncthreads();
        const uint32_t fi = cz_uni(CZX_CTL[CZX_C_FRAME]);
        if (fi >= a.n) break;
        const uint32_t f = a.frame_order ? cz_uni(a.frame_order[fi]) : fi;
        const uint64_t first = cz_uni64(a.frame_first[f]);
        if (first == 0 || first == CZX_DONE) continue;
#ifdef CZ_EMU_DEBUG
        if (threadIdx.x == 0) fprintf(stderr, "exec: frame %u start\n", f);
#endif
        const int rc = czx_run_frame(a, f, lit_scratch);
#ifdef CZ_EMU_DEBUG
        if ((threadIdx.x & 63) == 0) fprintf(stderr, "exec: frame %u wave %d rc %d\n", f, WAVE, rc);
#endif
        __syncthreads();
        if (threadIdx.x == 0) a.frame_first[f] = rc == 0 ? CZX_DONE : 0ull;   /* 0: cz_decode_frames_kernel decodes the frame from scratch */
    }
}
