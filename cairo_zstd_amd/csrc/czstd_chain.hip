/*
 * czstd_chain.hip — cz_chain_kernel: the FSE-chain pre-pass.
 *
 * The interleaved LL/OF/ML FSE state machines of a sequences section are ONE serial dependency
 * chain per block (sequence_section_decoder.cairo:223-286), and a chain step costs one LDS round
 * trip plus ~60 in-order instructions whatever the number of active lanes.  In
 * cz_decode_frames_kernel that chain runs on lane 0 of a 64-lane wave (1/64 of the issue
 * bandwidth used) and the frames in flight per CU are capped by that kernel's 10.6 KB of LDS.
 * Here the chain is all a lane does: CZC_SLOTS (12) frames per wave, one per lane, each with its own
 * 16-bit decoding tables (2.5 KB) and a 256-byte bit ring in LDS: 3 KB per chain, 48 chains per CU in
 * four waves, one per SIMD (the number of chains in flight divided by the step latency is this
 * kernel's throughput; 8, 16 and 21 slots per wave and two interleaved chains per lane measured
 * slower).  The other lanes only help staging bytes.  Per sequence the lane appends one 8-byte record
 * (the 32 stream bits that hold the extra bits | LL state, ML state, OF code) to the chain arena, and
 * per block the state->code maps of the LL and ML tables; cz_decode_frames_kernel then extracts the
 * extra bits, resolves offsets and executes the sequences without tables, bitstream or chain.
 *
 * This pass is a pure accelerator for well-formed frames: on ANY irregularity (malformed header,
 * table error, invalid code, overrun, left-over bits, more than 32 extra bits in a sequence,
 * arena overflow, > 64 symbols in a table description) — and for frames whose first sequences
 * section is short (chain_min_nseq), where it would not pay — it marks the whole frame "no chain info"
 * (frame_first[f] = 0) and the main kernel decodes that frame entirely by itself, producing the
 * reference's status codes in the reference's order.  Nothing here reports errors.
 */
#ifndef CZC_SLOTS
#define CZC_SLOTS 12
#endif
#define CZC_LPS (64 / CZC_SLOTS)   /* helper lanes per slot for staging */
static_assert(CZC_LPS >= 3, "three lanes of a slot build its three tables");
#define CZC_MAXSYM 64
#define CZC_RING 512u       /* fits in the build scratch it is overlaid with; filled at most 256 bytes per top-up */
#ifndef CZC_STEPS
#define CZC_STEPS 32u
#endif
static_assert(((CZC_STEPS * 58u + 7u) / 8u + 12u) + 12u <= 256u + 12u, "a top-up adds at most 256 bytes: one must be enough for the next group");
#define CZC_NEED ((CZC_STEPS * 58u + 7u) / 8u + 12u)   /* CZC_STEPS steps x 58 bits (32 extra bits + 26 state bits at most) + the 8 bytes a step reads below its cursor */
#define CZC_MAP_WORDS CZ_CHAIN_MAP_WORDS  /* per block in the arena: state -> code maps, 512 B LL + 512 B ML */
/* args.chain_min_nseq (default 2048): frames whose first sequences section is smaller are left to the
   main kernel — the pre-pass only pays for long chains (measured on the corpus-like mix). */

/* Chain-time decoding tables are 16 bits per state so that more chains fit in a CU's LDS (the number
 * of chains in flight is what bounds this kernel).  An entry holds only what the serial core needs:
 *   [15:6] v = (1 << (9 - num_bits)) | (base_line >> num_bits)     [5:1] extra bits of the code
 * base_line is always a multiple of 2^num_bits (fse_decoder.cairo:231-255: it is a multiple of the
 * slice width), so num_bits = clz10(v) and next state = ((v << num_bits) | bits) & 511.  The symbol
 * itself is not stored: the record carries the STATE and cz_decode_frames_kernel maps it to the code
 * with the per-block state->code byte maps this kernel leaves in the arena. */
#define CZC_E16(nb, base, xb) ((uint16_t)(((((1u << (9u - (nb))) | ((base) >> (nb))) << 6) | ((xb) << 1))))

struct CzChainSlot {
    uint16_t t_ll[512], t_ml[512], t_of[256];
    union {                                              /* table-build time | chain time */
        struct {
            __attribute__((aligned(16))) uint8_t stage[256]; /* head of the sequences section, linear; once the descriptions are
                                                                read: the symbol counters of the LL and OF table builds */
            int16_t probs[3][CZC_MAXSYM];                    /* normalised counts of LL, OF, ML */
            uint16_t counters_ml[CZC_MAXSYM];
        };
        struct {
            __attribute__((aligned(16))) uint8_t mirror[16];   /* mirror[8..15] == the last 8 bytes of ring */
            uint8_t ring[CZC_RING];                      /* reversed bitstream, indexed by absolute address & (CZC_RING - 1) */
        };
    };
};
struct CzChainShared { CzChainSlot slot[CZC_SLOTS]; uint32_t llml[96]; };

/* build_decoding_table (fse_decoder.cairo:156-256) into 16-bit chain entries + the state->code map
 * (global, bytes).  kind 0 LL, 1 OF, 2 ML.  Returns 1 if the table holds a code the sequence decoder
 * rejects (LL >= 36, OF >= 32, ML >= 53): such frames are left to the main kernel. */
__device__ static __attribute__((noinline)) int czc_fse_build16(uint16_t* table, const int16_t* probs, uint32_t nprobs, uint32_t log,
                                                                uint16_t* counters, const uint32_t* llml, uint32_t kind, uint8_t* map) {
    const uint32_t size = 1u << log, lim = kind == 0 ? 36u : (kind == 1 ? 32u : 53u);
    uint32_t neg = size; int bad = 0;
    for (uint32_t s = 0; s < nprobs; s++) {                             /* :169-188 */
        counters[s] = 0;
        if (probs[s] != 0 && s >= lim) bad = 1;
        if (probs[s] == -1) { neg--; table[neg] = (uint16_t)s; }
    }
    if (bad) return 1;
    uint32_t pos = 0; const uint32_t step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    for (uint32_t s = 0; s < nprobs; s++) {                             /* :190-226 */
        const int32_t p = probs[s];
        for (int32_t j = 0; j < p; j++) {
            table[pos] = (uint16_t)s;
            do { pos = (pos + step) & mask; } while (pos >= neg);
        }
    }
    for (uint32_t i = 0; i < size; i++) {                               /* :231-255, :377-400 */
        const uint32_t s = table[i];
        const uint32_t xb = kind == 1 ? s : (llml[(kind == 2 ? 40u : 0u) + s] >> 24);
        uint32_t nb, bl;
        if (i >= neg) { nb = log; bl = 0; }
        else {
            const uint32_t n = (uint32_t)probs[s], k = counters[s];
            counters[s] = (uint16_t)(k + 1);
            const uint32_t m = 1u << (cz_hbs(n) - 1), slices = (m == n) ? n : m * 2;
            const uint32_t dbl = slices - n, single = n - dbl, width = size / slices;
            nb = cz_hbs(width) - 1;
            if (k < dbl) { bl = single * width + k * width * 2; nb += 1; }
            else bl = (k - dbl) * width;
        }
        table[i] = CZC_E16(nb, bl, xb);
        if (map) map[i] = (uint8_t)s;
    }
    return 0;
}

/* 16 bytes at absolute address a, zero outside [S, E) */
__device__ static inline uint4 czc_load16(uintptr_t a, uintptr_t S, uintptr_t E) {
    uint4 v;
    if (a >= S && a + 16 <= E) { __builtin_memcpy(&v, (CZ_GLOBAL const void*)a, 16); return v; }
    if (a + 16 <= S || a >= E) { v.x = v.y = v.z = v.w = 0; return v; }
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t b = 0; b < 16; b++) { const uintptr_t q = a + b; if (q >= S && q < E) w[b >> 2] |= (uint32_t)(*(cz_gcptr)q) << (8 * (b & 3)); }
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
}
/* Ring top-up.  The 256-byte ring of slot k holds stream bytes [lo_k, lo_k + 256), indexed by absolute
 * address & 255.  Lanes k*LPS .. k*LPS+LPS-1 extend it downwards from old_lo to new_lo (both 16-aligned,
 * old_lo - new_lo <= 256) with 16-byte loads, all issued before the first LDS write.  Sk/Ek are the
 * helper lane's copy of its slot's stream bounds; old_lo/new_lo are the OWNER lane's values.
 * A wave stalls on global-memory latency here, so the caller tops up EVERY slot to the brim whenever
 * any slot runs low: one stall per ~50 chain steps instead of one per slot per block. */
#define CZC_PF ((16 + CZC_LPS - 1) / CZC_LPS)
__device__ static inline void czc_topup(CzChainShared& cs, intptr_t old_lo, intptr_t new_lo, uintptr_t Sk, uintptr_t Ek) {
    const int helper = (uint32_t)LANE / CZC_LPS < CZC_SLOTS;
    const uint32_t k = helper ? (uint32_t)LANE / CZC_LPS : 0, j = (uint32_t)LANE % CZC_LPS;
    const uintptr_t ok = ((uintptr_t)__shfl((uint32_t)((uint64_t)old_lo >> 32), (int)k) << 32) | __shfl((uint32_t)old_lo, (int)k);
    const uint32_t cnt_k = (uint32_t)__shfl((uint32_t)(old_lo - new_lo), (int)k) >> 4;       /* every lane takes part in the shuffle */
    const uint32_t cnt = helper ? cnt_k : 0;
    uint4 v[CZC_PF];
#pragma unroll
    for (uint32_t r = 0; r < CZC_PF; r++) { const uint32_t c = j + r * CZC_LPS; if (c < cnt) v[r] = czc_load16(ok - 16u * (c + 1), Sk, Ek); }
#pragma unroll
    for (uint32_t r = 0; r < CZC_PF; r++) {
        const uint32_t c = j + r * CZC_LPS;
        if (c < cnt) {
            const uint32_t slot = (uint32_t)((ok - 16u * (c + 1)) & (CZC_RING - 1));
            *(uint4*)&cs.slot[k].ring[slot] = v[r];
            if (slot == CZC_RING - 16) { *(uint32_t*)&cs.slot[k].mirror[8] = v[r].z; *(uint32_t*)&cs.slot[k].mirror[12] = v[r].w; }
        }
    }
}
/* 64 stream bits below ring-space bit address u (exclusive); see cz_ring_window */
__device__ static inline uint64_t czc_window(const CzChainSlot& sl, int32_t u) {
    const uint32_t ba = ((uint32_t)u >> 3) & (CZC_RING - 4), ph = (uint32_t)u & 31;
    const uint32_t w2 = *(const uint32_t*)(sl.ring + ba), w1 = *(const uint32_t*)(sl.ring + ba - 4), w0 = *(const uint32_t*)(sl.ring + ba - 8);
    /* bits [u-64, u): u = 32*wi + ph, word wi holds bits >= u when ph == 0 */
    const uint64_t hi = (((uint64_t)w2 << 32) | w1), lo = (((uint64_t)w1 << 32) | w0);
    const uint32_t h = ph ? (uint32_t)(hi >> ph) : w1, l = ph ? (uint32_t)(lo >> ph) : w0;
    return ((uint64_t)h << 32) | l;
}

/* `steps` chain steps of one lane (sequence_section_decoder.cairo:223-286, serial core).
 * TAIL: the group may contain the block's last sequence (which updates no state, :258).
 * Record per sequence: low word = the 32 stream bits below the cursor (they hold the sequence's
 * extra bits, read first, :239-256), high word = LL state | ML state << 9 | OF code << 18. */
template <bool TAIL>
__device__ static inline void czc_group(const CzChainSlot& sl, uint64_t* rec, uint32_t steps, uint32_t nseq, uint32_t done, uint32_t sbits,
                                        int32_t& u, uint32_t& sLL, uint32_t& sOF, uint32_t& sML, uint32_t& slow) {
    auto step = [&](uint32_t i) {
        const uint32_t ba = ((uint32_t)u >> 3) & (CZC_RING - 4);
        const uint32_t w2 = *(const uint32_t*)(sl.ring + ba), w1 = *(const uint32_t*)(sl.ring + ba - 4), w0 = *(const uint32_t*)(sl.ring + ba - 8);
        const uint32_t eLL = sl.t_ll[sLL], eOF = sl.t_of[sOF], eML = sl.t_ml[sML];
        const uint32_t xl = (eLL >> 1) & 31, xm = (eML >> 1) & 31, xo = (eOF >> 1) & 31, a_ = xl + xm + xo;
        const uint32_t ph = (uint32_t)u & 31;
#ifndef CZC_EXP_NOSTORE
        rec[done + i] = (uint64_t)__builtin_amdgcn_alignbit(w2, w1, ph) | ((uint64_t)(sLL | (sML << 9) | (xo << 18)) << 32);
#else
        if (u == 0x7FFFFFF) rec[done + i] = (uint64_t)__builtin_amdgcn_alignbit(w2, w1, ph) | ((uint64_t)(sLL | (sML << 9) | (xo << 18)) << 32);
#endif
        slow = slow > a_ ? slow : a_;                                   /* more than 32 extra bits: checked after the chain */
        const uint32_t sel = ph >= a_;
        const uint32_t xh = __builtin_amdgcn_alignbit(sel ? w2 : w1, sel ? w1 : w0, (ph - a_) & 31);
        const uint32_t vl = eLL >> 6, vm = eML >> 6, vo = eOF >> 6;
        const uint32_t nl = (uint32_t)__builtin_clz(vl) - 22, nm_ = (uint32_t)__builtin_clz(vm) - 22, no = (uint32_t)__builtin_clz(vo) - 22;   /* v != 0 in a built table */
        sLL = ((vl << nl) | __builtin_amdgcn_ubfe(xh, 32 - nl, nl)) & 511;       /* update order LL, ML, OF (:258-277) */
        sML = ((vm << nm_) | __builtin_amdgcn_ubfe(xh, 32 - nl - nm_, nm_)) & 511;
        sOF = ((vo << no) | __builtin_amdgcn_ubfe(xh, 32 - nl - nm_ - no, no)) & 255;
        if (TAIL) u -= (int32_t)((done + i + 1 == nseq) ? a_ : a_ + nl + nm_ + no);
        else u -= (int32_t)(a_ + nl + nm_ + no);
    };
    if (TAIL) { for (uint32_t i = 0; i < steps; i++) step(i); }
    else {
#pragma unroll
        for (uint32_t i = 0; i < CZC_STEPS; i++) step(i);
    }
}

/* Diagnostic build only: wave-level s_memtime sums per phase of this kernel, in args.prof[32..39]:
 * 32 block parse + table build, 33 ring fill + state init, 34 top-up events, 35 chain groups, 36 finalize,
 * 37 number of top-up events, 38 number of groups. */
#ifdef CZ_PROFILE
#define CZC_PROF_ACC(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); cprof[i] += n_ - ct_; ct_ = n_; } while (0)
#define CZC_PROF_CNT(i) do { cprof[i] += 1; } while (0)
#else
#define CZC_PROF_ACC(i) do { } while (0)
#define CZC_PROF_CNT(i) do { } while (0)
#endif
extern "C" __global__ void __launch_bounds__(CZ_WG_THREADS, 1) cz_chain_kernel(cz_batch_args a) {
    __shared__ CzChainShared cs;
    for (uint32_t i = (uint32_t)LANE; i < 36; i += 64) cs.llml[i] = CZ_LL_BASE[i] | ((uint32_t)CZ_LL_BITS[i] << 24);
    for (uint32_t i = (uint32_t)LANE; i < 53; i += 64) cs.llml[40 + i] = CZ_ML_BASE[i] | ((uint32_t)CZ_ML_BITS[i] << 24);
    __syncthreads();
    const int owner = LANE < CZC_SLOTS;
    CzChainSlot& sl = cs.slot[owner ? LANE : 0];
#ifdef CZ_PROFILE
    unsigned long long cprof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ct_ = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        /* ---- one frame per slot */
        uint32_t f = 0xFFFFFFFFu;
        if (owner) f = atomicAdd(a.chain_counter, 1u);
        int frame_live = owner && f < a.n;
        if (!__ballot(frame_live)) break;
        const uint8_t* src = nullptr; uint64_t len = 0, pos = 0;
        int punt = 0;
        uint32_t logs[3] = {0, 0, 0}; int32_t rles[3] = {-1, -1, -1};       /* carried across the frame's blocks (Repeat mode) */
        uint64_t first_hdr = 0, prev_hdr = 0;
        if (frame_live) {
            src = a.in_base + a.in_off[f]; len = a.in_len[f];
            /* frame header (frame.cairo:152-284): only its length and validity matter here */
            if (len < 5) punt = 1;
            else {
                const uint32_t magic = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
                const uint32_t d = src[4];
                const uint32_t single = (d >> 5) & 1, didf = d & 3, dl = didf == 3 ? 4 : didf, flag = d >> 6;
                const uint32_t fl = flag == 0 ? (single ? 1u : 0u) : flag == 1 ? 2u : flag == 2 ? 4u : 8u;
                const uint32_t hl = 5 + (single ? 0 : 1) + dl + fl;
                if (magic != 0xFD2FB528u || len < hl) punt = 1;
                else if (!single) { const uint32_t wd = src[5]; const uint64_t base = 1ull << (10 + (wd >> 3)); if (base + (base / 8) * (wd & 7) >= 4123168604160ull) punt = 1; }
                pos = hl;
            }
            if (punt) frame_live = 0;
        }
        int frame_done = !frame_live;
        /* ---- blocks: every slot advances to its next block that has sequences */
        while (__ballot(!frame_done)) {
            const uint8_t* blk = nullptr; uint32_t bsize = 0, nseq = 0, modes = 0, sbody = 0, blast = 0;
            int have = 0;
            if (!frame_done) {
                for (;;) {                                              /* block_decoder.cairo:237-321 */
                    if (len - pos < 3) { punt = 1; break; }
                    const uint32_t b0 = src[pos], b1 = src[pos + 1], b2 = src[pos + 2];
                    const uint32_t type = (b0 >> 1) & 3, size = (b0 >> 3) | (b1 << 5) | (b2 << 13);
                    blast = b0 & 1;
                    if (type == 3 || size > 128u * 1024u) { punt = 1; break; }
                    const uint64_t body = pos + 3; const uint32_t content = type == 1 ? 1u : size;
                    if (len - body < content) { punt = 1; break; }
                    if (type != 2) { pos = body + content; if (blast) break; continue; }
                    /* literals section header (literals_section.cairo:81-175): sizes only */
                    const uint8_t* p = src + body;
                    if (size == 0) { punt = 1; break; }
                    const uint32_t l0 = p[0], lt = l0 & 3, fmt = (l0 >> 2) & 3;
                    const uint32_t need = lt <= 1 ? ((fmt == 0 || fmt == 2) ? 1u : (fmt == 1 ? 2u : 3u)) : (fmt <= 1 ? 3u : (fmt == 2 ? 4u : 5u));
                    if (size < need) { punt = 1; break; }
                    const uint32_t l1 = need > 1 ? p[1] : 0, l2 = need > 2 ? p[2] : 0, l3 = need > 3 ? p[3] : 0, l4 = need > 4 ? p[4] : 0;
                    uint32_t upper;
                    if (lt <= 1) { const uint32_t regen = (fmt == 0 || fmt == 2) ? l0 >> 3 : (fmt == 1 ? (l0 >> 4) + (l1 << 4) : (l0 >> 4) + (l1 << 4) + (l2 << 12)); upper = lt == 1 ? 1u : regen; }
                    else upper = fmt <= 1 ? (l1 >> 6) + (l2 << 2) : (fmt == 2 ? (l2 >> 2) + (l3 << 6) : (l2 >> 6) + (l3 << 2) + (l4 << 10));
                    if (size - need < upper) { punt = 1; break; }
                    const uint32_t so = need + upper, sl_ = size - so;  /* sequence_section.cairo:77-114 */
                    if (sl_ == 0) { punt = 1; break; }
                    const uint32_t s0 = p[so];
                    if (s0 == 0) { pos = body + content; if (blast) break; continue; }
                    uint32_t n = 0, hb = 0;
                    if (s0 <= 127) { if (sl_ < 2) { punt = 1; break; } n = s0; hb = 1; }
                    else if (s0 <= 254) { if (sl_ < 3) { punt = 1; break; } n = ((s0 - 128) << 8) + p[so + 1]; hb = 2; }
                    else { if (sl_ < 4) { punt = 1; break; } n = p[so + 1] + ((uint32_t)p[so + 2] << 8) + 0x7F00u; hb = 3; }
                    if (n == 0) { pos = body + content; if (blast) break; continue; }   /* 128,0: sequences = 0 but a modes byte */
                    if (first_hdr == 0 && n < a.chain_min_nseq) { punt = 1; break; }
                    blk = p; bsize = size; nseq = n; modes = p[so + hb]; sbody = so + hb + 1;
                    pos = body + content; have = 1;
                    break;
                }
                if (punt || !have) frame_done = 1;
            }
            /* ---- stage the head of the sequences section (table descriptions) linearly: 256 bytes */
            {
                const unsigned long long hm = __ballot(have);
                if (hm) {
                    const uintptr_t base = (uintptr_t)blk + sbody, Sx = (uintptr_t)blk, Ex = (uintptr_t)blk + bsize;
                    const int helper = (uint32_t)LANE / CZC_LPS < CZC_SLOTS;
                    const uint32_t k = helper ? (uint32_t)LANE / CZC_LPS : 0, j = (uint32_t)LANE % CZC_LPS;
                    const uintptr_t bk = ((uintptr_t)__shfl((uint32_t)((uint64_t)base >> 32), (int)k) << 32) | __shfl((uint32_t)base, (int)k);
                    const uintptr_t Sk = ((uintptr_t)__shfl((uint32_t)((uint64_t)Sx >> 32), (int)k) << 32) | __shfl((uint32_t)Sx, (int)k);
                    const uintptr_t Ek = ((uintptr_t)__shfl((uint32_t)((uint64_t)Ex >> 32), (int)k) << 32) | __shfl((uint32_t)Ex, (int)k);
                    if (helper && ((hm >> k) & 1ull)) for (uint32_t c = j; c < 16; c += CZC_LPS) *(uint4*)&cs.slot[k].stage[16 * c] = czc_load16(bk + 16 * c, Sk, Ek);
                    __syncthreads();
                }
            }
            /* ---- tables (sequence_section_decoder.cairo:405-647), serial per lane */
            uint32_t bitoff = 0, mapflags = 0; uint64_t hdr = 0;
            if (have) {                                                 /* arena: 4-word header + code maps + nseq records */
                const unsigned long long units = 4ull + CZC_MAP_WORDS + nseq;
                hdr = 8ull + atomicAdd(a.chain_top, units);          /* indices 0..7 are reserved (0 = none) */
                if (hdr + units > a.chain_capacity) { punt = 1; have = 0; frame_done = 1; }
            }
            uint32_t binfo = 0;                                         /* per table, 10 bits: (symbols - 1) | log << 6; log 0 = nothing to build */
            if (have) {
                uint32_t off = sbody;
                uint8_t* maps = (uint8_t*)(a.chain_arena + hdr + 4);
                for (int t = 0; t < 3 && !punt; t++) {
                    uint16_t* table = t == 0 ? sl.t_ll : (t == 1 ? sl.t_of : sl.t_ml);
                    uint8_t* map = t == 0 ? maps : (t == 2 ? maps + 512 : nullptr);   /* the OF code travels in the record */
                    const uint32_t md = (modes >> (6 - 2 * t)) & 3, max_log = t == 1 ? 8u : 9u;
                    if (md != 3) mapflags |= 1u << t;
                    if (md == 0) {
                        const int8_t* d = t == 0 ? CZ_LL_DEFAULT : t == 1 ? CZ_OF_DEFAULT : CZ_ML_DEFAULT;
                        const uint32_t n = t == 0 ? 36u : t == 1 ? 29u : 53u, lg = t == 1 ? 5u : 6u;
                        for (uint32_t s = 0; s < n; s++) sl.probs[t][s] = d[s];
                        binfo |= ((n - 1) | (lg << 6)) << (10 * t);
                        logs[t] = lg; rles[t] = -1;
                    } else if (md == 1) {
                        /* RLE: a one-state table (num_bits 0, base 0) at state 0 (:437-446) */
                        if (off >= bsize) { punt = 1; break; }
                        const uint32_t sym = blk[off]; off += 1;
                        if (sym >= (t == 0 ? 36u : (t == 1 ? 32u : 53u))) { punt = 1; break; }
                        table[0] = CZC_E16(0u, 0u, t == 1 ? sym : (cs.llml[(t == 2 ? 40u : 0u) + sym] >> 24));
                        if (map) map[0] = (uint8_t)sym;
                        rles[t] = (int32_t)sym;
                    } else if (md == 2) {
                        CzFBits br; br.g = (cz_gcptr)(blk + off); br.len = bsize - off; br.idx = 0; br.stage = sl.stage; br.stage_lo = 0; br.stage_hi = 0;
                        if (off - sbody < 256) { br.stage = sl.stage + (off - sbody); br.stage_hi = 256 - (off - sbody); }
                        uint32_t np, lg, used;
                        if (cz_fse_read_probs(br, max_log, sl.probs[t], &np, &lg, &used, 100, CZC_MAXSYM) || np > CZC_MAXSYM) { punt = 1; break; }
                        if (np == 0) { punt = 1; break; }
                        binfo |= ((np - 1) | (lg << 6)) << (10 * t);
                        logs[t] = lg; rles[t] = -1; off += used;
                        if (off > bsize) { punt = 1; break; }
                    } else if (rles[t] < 0 && logs[t] == 0) { punt = 1; break; }     /* Repeat of nothing */
                }
                bitoff = off;
                if (punt) { have = 0; frame_done = 1; }
            }
            /* ---- build the tables: the LL, OF and ML table of a slot on three different lanes, side by side */
            {
                __syncthreads();                                        /* descriptions read: `stage` may become counters */
                const uint32_t k = (uint32_t)LANE / CZC_LPS < CZC_SLOTS ? (uint32_t)LANE / CZC_LPS : 0, t = (uint32_t)LANE % CZC_LPS;
                const int role = (uint32_t)LANE / CZC_LPS < CZC_SLOTS && t < 3;
                const uint32_t info = (__shfl(have ? binfo : 0u, (int)k) >> (10 * (t < 3 ? t : 0))) & 0x3FFu;
                const uint64_t hk = ((uint64_t)__shfl((uint32_t)(hdr >> 32), (int)k) << 32) | __shfl((uint32_t)hdr, (int)k);
                int bad = 0;
                if (role && (info >> 6)) {
                    CzChainSlot& sk = cs.slot[k];
                    uint16_t* table = t == 0 ? sk.t_ll : (t == 1 ? sk.t_of : sk.t_ml);
                    uint16_t* counters = t == 0 ? (uint16_t*)sk.stage : (t == 1 ? (uint16_t*)sk.stage + CZC_MAXSYM : sk.counters_ml);
                    uint8_t* maps = (uint8_t*)(a.chain_arena + hk + 4);
                    uint8_t* map = t == 0 ? maps : (t == 2 ? maps + 512 : nullptr);
                    bad = czc_fse_build16(table, sk.probs[t], (info & 63u) + 1u, info >> 6, counters, cs.llml, t, map);
                }
                __syncthreads();
                const int base = LANE < CZC_SLOTS ? LANE * (int)CZC_LPS : 0;
                const int anybad = __shfl(bad, base) | __shfl(bad, base + 1) | __shfl(bad, base + 2);
                if (have && anybad) { punt = 1; have = 0; frame_done = 1; }
            }
            CZC_PROF_ACC(0);
            /* ---- bit ring: fill every live slot's ring with the top 256 bytes of its stream */
            const uintptr_t S = have ? (uintptr_t)blk + bitoff : 0, E = have ? (uintptr_t)blk + bsize : 0;
            const uint32_t sbits = (uint32_t)(S & (CZC_RING - 1)) * 8u;
            const intptr_t ring_base = (intptr_t)(S & ~(uintptr_t)(CZC_RING - 1));          /* ring-space bit u <-> byte ring_base + (u >> 3) */
            intptr_t loaded_lo = 0;
            uintptr_t Sk, Ek;                                           /* helper lanes: bounds of their slot's stream */
            {
                const int helper = (uint32_t)LANE / CZC_LPS < CZC_SLOTS;
                const int k = helper ? (int)((uint32_t)LANE / CZC_LPS) : 0;
                Sk = ((uintptr_t)__shfl((uint32_t)((uint64_t)S >> 32), k) << 32) | __shfl((uint32_t)S, k);
                Ek = ((uintptr_t)__shfl((uint32_t)((uint64_t)E >> 32), k) << 32) | __shfl((uint32_t)E, k);
                const unsigned long long hm = __ballot(have);
                __syncthreads();
                if (hm) {
                    const intptr_t hi = (intptr_t)((E + 15) & ~(uintptr_t)15);
                    loaded_lo = have ? hi - 256 : 0;
                    czc_topup(cs, have ? hi : 0, loaded_lo, Sk, Ek);
                    __syncthreads();
                }
            }
            /* ---- chain */
            int32_t u = 0; uint32_t sLL = 0, sOF = 0, sML = 0, done = 0, slow = 0;
            int chain_live = have;
            if (have) {
                int32_t p = (int32_t)(E - S) * 8; int skipped = 0;
                for (;;) {                                              /* padding :46-64 */
                    const uint32_t b = p > 0 ? (uint32_t)(czc_window(sl, (int32_t)sbits + p) >> 63) : 0; p -= 1; skipped++;
                    if (b == 1 || skipped > 8) break;
                }
                if (skipped > 8) slow = 64;
                uint32_t stv[3] = {0, 0, 0};
                for (int t = 0; t < 3; t++) {                           /* init order LL, OF, ML (:207-218) */
                    if (rles[t] >= 0) continue;
                    stv[t] = p > 0 ? (uint32_t)(czc_window(sl, (int32_t)sbits + p) >> (64 - logs[t])) : 0; p -= (int32_t)logs[t];
                }
                sLL = stv[0]; sOF = stv[1]; sML = stv[2];
                if (p < 0) slow = 64;
                u = (int32_t)sbits + p;
            }
            uint64_t* rec = a.chain_arena + hdr + 4 + CZC_MAP_WORDS;
            CZC_PROF_ACC(1);
            while (__ballot(chain_live)) {
                /* keep CZC_NEED bytes below every live cursor staged; when one slot runs low, all top up */
                {
                    const intptr_t curb = ring_base + ((u > 0 ? u - 1 : 0) >> 3);       /* byte that holds the next unread bit */
                    const int need = chain_live && (curb - (intptr_t)CZC_NEED < loaded_lo);
                    if (__ballot(need)) {
                        /* lowest start whose 256 bytes still cover the word at the cursor */
                        intptr_t new_lo = chain_live ? ((curb - (intptr_t)(CZC_RING - 4) + 15) & ~(intptr_t)15) : loaded_lo;
                        if (new_lo > loaded_lo) new_lo = loaded_lo;
                        if (new_lo < loaded_lo - 256) new_lo = loaded_lo - 256;      /* czc_topup moves at most 16 pieces */
                        __syncthreads();
                        czc_topup(cs, loaded_lo, new_lo, Sk, Ek);
                        loaded_lo = new_lo;
                        __syncthreads();
                        CZC_PROF_ACC(2); CZC_PROF_CNT(5);
                    }
                }
                {
                    /* uniform choice of the loop flavour for this group of CZC_STEPS steps */
                    const uint32_t left = nseq - done;
                    const int tail = __ballot(chain_live && left <= CZC_STEPS) != 0;
                    if (chain_live) {
                        const uint32_t steps = left < CZC_STEPS ? left : CZC_STEPS;
                        if (!tail) czc_group<false>(sl, rec, CZC_STEPS, nseq, done, sbits, u, sLL, sOF, sML, slow);
                        else czc_group<true>(sl, rec, steps, nseq, done, sbits, u, sLL, sOF, sML, slow);
                        done += steps;
                        if (done >= nseq) chain_live = 0;
                    }
                    CZC_PROF_ACC(3); CZC_PROF_CNT(6);
                }
            }
            /* ---- finalize the block */
            if (have) {
                /* the cursor only moves down, so an overrun (NotEnoughBytes, :281) shows in its final value */
                if (slow > 32 || u - (int32_t)sbits != 0) { punt = 1; frame_done = 1; }   /* > 32 extra bits / overrun / ExtraBits */
                else {
                    uint64_t* h = a.chain_arena + hdr;
                    h[0] = ((uint64_t)nseq << 32) | mapflags; h[1] = bitoff; h[2] = 0; h[3] = 0;
                    if (prev_hdr) a.chain_arena[prev_hdr + 2] = hdr; else first_hdr = hdr;
                    prev_hdr = hdr;
                    if (blast) frame_done = 1;
                }
            } else if (!frame_done && blast) frame_done = 1;
            __syncthreads();
        }
        if (owner && f < a.n) a.frame_first[f] = punt ? 0 : first_hdr;
        CZC_PROF_ACC(4);
    }
#ifdef CZ_PROFILE
    if (LANE == 0 && a.prof) for (int i = 0; i < 8; i++) atomicAdd(&a.prof[32 + i], cprof[i]);
#endif
}
