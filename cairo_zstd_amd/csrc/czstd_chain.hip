/*
 * czstd_chain.hip — cz_scan_kernel + cz_chain_kernel: the block-parallel FSE-chain pre-pass.
 *
 * The interleaved LL/OF/ML FSE state machines of a sequences section are ONE serial dependency
 * chain per block (sequence_section_decoder.cairo:223-286).  With every block of a batch in flight
 * at once the pass lasts (sequences per block) x (latency of one chain step), so the step is
 * written for latency, one wave per SIMD; the unit of work is a BLOCK (cz_scan_kernel lists the blocks of
 * all frames, a slot takes the next one when its block is done):
 *   - a chain is spread over a QUAD of lanes: lane 0 runs the LL state, lane 1 the ML state, lane 2
 *     the OF state (update order LL, ML, OF, :258-277), lane 3 idles on a dummy table.  Each lane does
 *     ONE table lookup, one v_ffbh and one v_bfe per step; the bit offsets of the three fields and
 *     the total come from four quad_perm DPP operations; the cursor advances with one v_dot4c;
 *     tables of a refilled slot are built by the whole wave (czc_fse_build_wave).
 *   - CZC_SLOTS (10) quads per wave, four waves per CU: 40 chains per CU (LDS: 16-bit
 *     decoding tables 2.5 KB + a 512-byte bit ring per chain; 48 would fit, 40 leave 25 KB per CU to the huff0 waves (cz_huf1_kernel) that
 *     runs next to this kernel).
 *   - the 32 steps of a group are one hand-scheduled inline-asm block (czc_group_asm): the ring words
 *     of step i+1 are requested before the state bits of step i are extracted, the next table entry is
 *     loaded straight into the upper half of its register (ds_read_u16_d16_hi), and the record store,
 *     the cursor update and the record word of the next step sit in the shadow of that load.
 *     czc_step is the same step in plain C++ (tail groups, and the CPU emulator of tests/emu).
 * Per sequence the quad appends one 8-byte record (the 32 stream bits below the cursor, which begin
 * with the sequence's extra bits | LL state, ML state << 9, OF state << 18) to the chain arena, and per
 * block the state->code byte maps of the three tables; the decode kernels then extract the extra
 * bits, resolve offsets and execute the sequences without tables, bitstream or chain.
 *
 * This pass is a pure accelerator for well-formed frames: on ANY irregularity (malformed header,
 * table error, invalid code, overrun, left-over bits, arena overflow, > 64 symbols in a table description) — and for frames
 * whose first sequences section is shorter than chain_min_nseq (0 by default) — it marks the whole frame "no chain info"
 * (frame_first[f] = 0) and the main kernel decodes that frame entirely by itself, producing the
 * reference's status codes in the reference's order.  Nothing here reports errors.
 */
#ifndef CZC_SLOTS
#define CZC_SLOTS 10        /* 40 chains per CU = 10 240 on the chip; 12 also fit the LDS, but leave no room for cz_huf1_kernel next to this kernel */
#endif
#define CZC_LPS 4           /* lanes per slot: the quad */
static_assert(CZC_SLOTS * CZC_LPS <= 64, "quads of one wave");
#define CZC_MAXSYM 64
#define CZC_RING 512u       /* fits in the build scratch it is overlaid with; filled at most 256 bytes per top-up */
#define CZC_STEPS 32u       /* steps per group (the asm block is unrolled for exactly this many) */
static_assert(((CZC_STEPS * 58u + 7u) / 8u + 12u) + 12u <= 256u + 12u, "a top-up adds at most 256 bytes: one must be enough for the next group");
#define CZC_NEED ((CZC_STEPS * 58u + 7u) / 8u + 12u)   /* CZC_STEPS steps x 58 bits (32 extra bits + 26 state bits at most) + the 8 bytes a step reads below its cursor;
                                                          also covers CZC_WIDE_STEPS (16) steps of czc_step at 89 bits (63 extra bits) each */
#define CZC_MAP_WORDS CZ_CHAIN_MAP_WORDS  /* per block in the arena: state -> code maps, 512 B LL + 512 B ML + 256 B OF */
/* args.chain_min_nseq (default 0): frames whose first sequences section is smaller are left to the main kernel (with
   blocks as the unit of work short chains cost little; the knob stays for experiments). */

/* Chain-time decoding tables are 16 bits per state so that more chains fit in a CU's LDS.  An entry
 * holds only what the serial core needs:
 *   [15:6] v = (1 << (9 - num_bits)) | (base_line >> num_bits)     [4:0] extra bits of the code
 * base_line is always a multiple of 2^num_bits (fse_decoder.cairo:231-255: it is a multiple of the
 * slice width).  Loaded into the UPPER half of a register (E = entry << 16): num_bits = clz(E), and
 * next state = bits [30:22] of (E << num_bits) | the num_bits stream bits.  The symbol itself is not
 * stored: the record carries the STATE and the decode kernels map it to the code with the per-block
 * state->code byte maps this kernel leaves in the arena. */
#define CZC_E16(nb, base, xb) ((uint16_t)(((((1u << (9u - (nb))) | ((base) >> (nb))) << 6) | (xb))))
#define CZC_E16_IDLE 0x8000u   /* num_bits 0, base 0, no extra bits: what a lane without a chain spins on */

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
            __attribute__((aligned(16))) uint8_t mirror[16];   /* mirror[4..15] == the last 12 bytes of ring */
            uint8_t ring[CZC_RING];                      /* reversed bitstream, indexed by absolute address & (CZC_RING - 1) */
        };
    };
};
struct CzChainShared { CzChainSlot slot[CZC_SLOTS]; uint32_t llml[96]; uint16_t idle[2]; };

/* build_decoding_table (fse_decoder.cairo:156-256) into 16-bit chain entries + the state->code map
 * (global, bytes), by the whole wave: all 64 lanes call this together, and the tables of the refilled
 * slots are built one after the other.  kind 0 LL, 1 OF, 2 ML.  Returns 1 (uniform) if the table holds a
 * code the sequence decoder rejects (LL >= 36, OF >= 32, ML >= 53): such frames are left to the main kernel.
 * (One lane per table, three tables side by side, was measured at ~600 clocks per cell: every step of those
 * loops is a dependent LDS round trip on a lone wave.)
 *   A  lane = symbol: "less than one" symbols go to the top cells (:169-188), an exclusive scan of the positive counts gives
 *      every symbol the rank of its first cell in spreading order;
 *   B  symof[rank] = symbol by a prefix maximum over the marks symof[first rank of s] = s, then lane = step j of the
 *      spreading walk: cell (j * step) & mask, skipped when it is a top cell, takes the symbol of its rank among the
 *      cells not skipped (:190-226; the walk visits every cell exactly once because step is odd);
 *   C  lane = cell, 64 cells at a time in ascending order: k = how many lower cells hold the same symbol (lanes of equal
 *      symbol are matched with six ballots, earlier chunks are in counters[]), then the entry as above (:231-255).
 * symof: size bytes of scratch, counters: 64 halfwords. */
__device__ static inline __attribute__((always_inline)) int czc_fse_build_wave(uint16_t* table, const int16_t* probs, uint32_t nprobs, uint32_t log,
                                                                   uint8_t* symof, uint16_t* counters, const uint32_t* llml, uint32_t kind, cz_gptr map) {
    const uint32_t size = 1u << log, mask = size - 1, lim = kind == 0 ? 36u : (kind == 1 ? 32u : 53u);
    const uint32_t lane = (uint32_t)LANE;
    const unsigned long long lt = (1ull << lane) - 1ull;
    /* A */
    const int32_t p = lane < nprobs ? (int32_t)probs[lane] : 0;
    if (__ballot(p != 0 && lane >= lim)) return 1;
    const unsigned long long lowm = __ballot(p == -1);
    const uint32_t neg = size - (uint32_t)__popcll(lowm);
    if (p == -1) table[size - 1u - (uint32_t)__popcll(lowm & lt)] = (uint16_t)lane;
    counters[lane] = 0;
    const uint32_t cnt = p > 0 ? (uint32_t)p : 0u;
    const uint32_t cum = cz_wave_incl_scan(cnt) - cnt;
    for (uint32_t i = 4u * lane; i < size; i += 256u) *(uint32_t*)(symof + i) = 0u;
    cz_wave_sync();
    if (cnt && cum < size) symof[cum] = (uint8_t)lane;
    cz_wave_sync();
    /* B: prefix maximum, `per` consecutive bytes per lane */
    {
        const uint32_t per = size >= 64u ? size >> 6 : 1u, b0 = lane * per;
        const int act = b0 < size;
        uint32_t run = 0, vals[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) { if (act && j < per) { const uint32_t v = symof[b0 + j]; run = v > run ? v : run; } vals[j] = run; }
        uint32_t inc = act ? run : 0u;
#define CZC_MAX_STEP(CTRL, RM) do { const uint32_t o_ = cz_dpp<CTRL, RM>(0u, inc); inc = o_ > inc ? o_ : inc; } while (0)
        CZC_MAX_STEP(CZ_DPP_SHR1, 0xF); CZC_MAX_STEP(CZ_DPP_SHR2, 0xF); CZC_MAX_STEP(CZ_DPP_SHR4, 0xF); CZC_MAX_STEP(CZ_DPP_SHR8, 0xF);
        CZC_MAX_STEP(CZ_DPP_BCAST15, 0xA); CZC_MAX_STEP(CZ_DPP_BCAST31, 0xC);
#undef CZC_MAX_STEP
        const uint32_t exc = cz_dpp<CZ_DPP_WAVE_SHR1, 0xF>(0u, inc);
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) if (act && j < per) symof[b0 + j] = (uint8_t)(vals[j] > exc ? vals[j] : exc);
    }
    cz_wave_sync();
    {
        const uint32_t step = (size >> 1) + (size >> 3) + 3u;
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
    /* C */
    for (uint32_t i0 = 0; i0 < size; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const int in = i < size, cell = in && i < neg;
        const uint32_t s = in ? (uint32_t)table[i] : 0u;
        unsigned long long same = __ballot(cell);
#pragma unroll
        for (uint32_t b = 0; b < 6; b++) { const int bit = (int)((s >> b) & 1u); const unsigned long long m = __ballot(bit); same &= bit ? m : ~m; }
        const uint32_t n = cell ? (uint32_t)probs[s] : 1u;
        const uint32_t k = (cell ? (uint32_t)counters[s] : 0u) + (uint32_t)__popcll(same & lt);
        cz_wave_sync();                                                 /* every lane has read counters[] */
        if (cell && (same >> lane) == 1ull) counters[s] = (uint16_t)(k + 1u);   /* the highest lane of the group */
        const uint32_t xb = kind == 1 ? s : (llml[(kind == 2 ? 40u : 0u) + s] >> 24);
        uint32_t nb = log, bl = 0;
        if (cell) {
            const uint32_t m = 1u << (cz_hbs(n) - 1), slices = (m == n) ? n : m * 2;
            const uint32_t dbl = slices - n, single = n - dbl, width = size >> (cz_hbs(slices) - 1);
            nb = cz_hbs(width) - 1;
            if (k < dbl) { bl = single * width + k * width * 2; nb += 1; }
            else bl = (k - dbl) * width;
        }
        if (in) { table[i] = CZC_E16(nb, bl, xb); if (map) map[i] = (uint8_t)s; }
        cz_wave_sync();
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
/* Ring top-up.  The ring of a slot holds stream bytes [lo, lo + 512), indexed by absolute address & 511.
 * The four lanes of the slot's quad extend it downwards from old_lo to new_lo (both 16-aligned,
 * old_lo - new_lo <= 256).  Every lane of a quad passes the same arguments.  The 256 bytes below the
 * ring are always on their way or already in registers (czc_prefetch, issued right after the previous
 * top-up), so a top-up is sixteen LDS writes per slot and no wait for global memory; the caller tops up
 * EVERY slot whenever any slot runs low. */
#define CZC_PF (16 / CZC_LPS)
struct CzcPre { uint4 v[CZC_PF]; };
__device__ static inline void czc_prefetch(CzcPre& pre, int has_slot, intptr_t lo, uintptr_t Sk, uintptr_t Ek) {
    const uint32_t j = (uint32_t)LANE & 3u;
#pragma unroll
    for (uint32_t r = 0; r < CZC_PF; r++) {
        const uint32_t c = j + r * CZC_LPS;
        if (has_slot) pre.v[r] = czc_load16((uintptr_t)lo - 16u * (c + 1), Sk, Ek);
    }
}
__device__ static inline void czc_commit(CzChainSlot& sk, int has_slot, intptr_t old_lo, intptr_t new_lo, const CzcPre& pre) {
    const uint32_t j = (uint32_t)LANE & 3u;
    const uint32_t cnt = has_slot ? (uint32_t)(old_lo - new_lo) >> 4 : 0u;
#pragma unroll
    for (uint32_t r = 0; r < CZC_PF; r++) {
        const uint32_t c = j + r * CZC_LPS;
        if (c < cnt) {
            const uint32_t slot = (uint32_t)(((uintptr_t)old_lo - 16u * (c + 1)) & (CZC_RING - 1));
            *(uint4*)&sk.ring[slot] = pre.v[r];
            if (slot == CZC_RING - 16) { *(uint32_t*)&sk.mirror[4] = pre.v[r].y; *(uint32_t*)&sk.mirror[8] = pre.v[r].z; *(uint32_t*)&sk.mirror[12] = pre.v[r].w; }
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

/* ---- the chain step ----------------------------------------------------------------------------
 * Per-lane state of a chain.  The three working lanes of a quad hold the same cursor (u, ph, ring
 * words) and each its own table state. */
struct CzcLane {
    uint32_t E;              /* current table entry << 16 */
    uint32_t S;              /* current state of this lane's table */
    int32_t  u;              /* ring-space bit address just above the next unread bit */
    uint32_t ph;             /* u & 31 */
    uint32_t w0, w1, w2;     /* ring words at ((u >> 5) & 127) - 2, - 1, - 0 */
    uint32_t slow;           /* largest number of extra bits seen in one sequence of the current asm group (> 32: the group is redone by czc_step) */
};
struct CzcRole {
    const uint16_t* tb;      /* this lane's table (LDS) */
    const uint8_t* ringm8;   /* this slot's ring - 8 (LDS) */
    uint32_t m1, m2;         /* 0xFF00 where the LL / ML state bits precede this lane's in the stream */
    uint32_t sh;             /* position of this lane's state in the record's high word */
    uint32_t lane0;          /* lane 0 of its quad */
    uint32_t sbits;          /* ring-space bit address of the start of the block's bitstream (u - sbits = bits still unread) */
};
#define CZC_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int CTRL> __device__ static inline uint32_t czc_qp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

__device__ static inline void czc_ring_words(CzcLane& c, const CzcRole& ro) {
    const uint8_t* ba = ro.ringm8 + (__builtin_amdgcn_ubfe((uint32_t)c.u, 5, 7) << 2);
    c.w0 = *(const uint32_t*)ba; c.w1 = *(const uint32_t*)(ba + 4); c.w2 = *(const uint32_t*)(ba + 8);
    c.ph = (uint32_t)c.u & 31u;
}
/* 32 stream bits below ring-space bit address v, straight from the slot's ring */
__device__ static inline uint32_t czc_bits32(const CzcRole& ro, uint32_t v) {
    const uint8_t* ba = ro.ringm8 + (__builtin_amdgcn_ubfe(v, 5, 7) << 2);
    return __builtin_amdgcn_alignbit(*(const uint32_t*)(ba + 8), *(const uint32_t*)(ba + 4), v & 31u);
}
/* One step (sequence_section_decoder.cairo:223-286, serial core) in plain C++: all 64 lanes call it
 * together.  `act`: this lane's chain takes the step; `last`: it is the block's last sequence (no
 * state update, :258); `store`: this lane writes the record.  The asm block below is this, scheduled — for
 * sequences of at most 32 extra bits.  This version takes any sequence (up to 16 + 16 + 31 extra bits): the record
 * of a wider one carries, instead of the 32 stream bits, the bit position of the sequence (bit 31 of the high word
 * set), and the decode kernel reads the extra bits from the bitstream itself.  Callers keep
 * CZC_WIDE_STEPS x 89 bits below the cursor staged. */
#define CZC_WIDE_STEPS 16u
static_assert((CZC_WIDE_STEPS * 89u + 7u) / 8u + 12u <= CZC_NEED && CZC_STEPS % CZC_WIDE_STEPS == 0, "a redone asm group is whole wide groups, each within the staged bytes");
__device__ static inline void czc_step(CzcLane& c, const CzcRole& ro, CZ_GLOBAL uint64_t* rec, int act, int last, int store) {
    const uint32_t n = (uint32_t)__builtin_clz(c.E);                    /* E != 0 in a built table */
    const uint32_t x = __builtin_amdgcn_ubfe(c.E, 16, 5);
    const uint32_t p = (n << 8) + x;
    const uint32_t T = p + czc_qp<CZC_QP(1, 2, 0, 3)>(p) + czc_qp<CZC_QP(2, 0, 1, 3)>(p);   /* byte 0: extra bits of the sequence, byte 1: state bits */
    const uint32_t t1 = czc_qp<CZC_QP(0, 0, 0, 0)>(p) & ro.m1, t2 = czc_qp<CZC_QP(1, 1, 1, 1)>(p) & ro.m2;
    const uint32_t incl = (t1 + t2 + p) >> 8;                           /* state bits up to and including this lane's */
    const uint32_t a_ = T & 0xFFu;
    /* 32 stream bits that follow the a_ extra bits (read order: extras first, :239) */
    const uint32_t xh = czc_bits32(ro, (uint32_t)c.u - a_);
    const uint32_t bits = __builtin_amdgcn_ubfe(xh, (0u - incl) & 31u, n);
    const uint32_t window = czc_bits32(ro, (uint32_t)c.u);
    const uint32_t cS = c.S << ro.sh;
    const uint32_t H = cS + czc_qp<CZC_QP(1, 2, 0, 3)>(cS) + czc_qp<CZC_QP(2, 0, 1, 3)>(cS);
    if (act) {
        if (store) *rec = a_ <= 32 ? ((uint64_t)window | ((uint64_t)H << 32))
                                   : ((uint64_t)((uint32_t)c.u - ro.sbits) | ((uint64_t)(H | CZC_REC_WIDE) << 32));
        if (last) c.u -= (int32_t)a_;
        else {
            c.S = (((c.E >> 22) << n) | bits) & 511u;                  /* (v << num_bits) carries the marker to bit 9 */
            c.E = (uint32_t)ro.tb[c.S] << 16;
            c.u -= (int32_t)(a_ + (T >> 8));
        }
        czc_ring_words(c, ro);
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
/* CZC_STEPS steps of every chain of the wave, none of them a block's last sequence.  Lanes without a live chain spin on the
 * idle entry and store into the sink.  Registers v100..v123 are scratch.  Wait states: a VGPR written by a VALU instruction is
 * read through DPP no sooner than two instructions later; the result of v_dot4c is read no sooner than four
 * instructions later (gfx940-family dot hazard); vcc is read no sooner than two instructions after v_cmp; a data
 * register of a store of more than 8 bytes is not written by the instruction right behind the store.
 *   v100 n   v101 x << 16 | 64   v102 p   v103 v = E >> 22   v105 T   v106 t1   v107 t2   v108 a   v109 o
 *   v110 d   v111 hi  v112 lo  v113 xh  v114 wi   v115 ba  v116 bits  v117 A   v118 c   v[124:125] ring words 0, 1
 *   v[120:121] / v[122:123]  record (window, states) of the even / odd step, stored together after the odd one */
#define CZC_ASM_HEAD(WIN) \
    CZC_ASM_LGKM_E                                 /* the table entry; the ring words may still be on their way */ \
    "v_ffbh_u32 v100, %[E]\n" \
    "v_and_or_b32 v101, %[E], %[XM], %[K64]\n" \
    "v_sub_u32 v102, v101, v100\n" \
    "v_lshrrev_b32 v103, 22, %[E]\n" \
    "v_and_b32 %[PH], 31, %[U]\n"                 /* phase of the cursor (and the second instruction between v102 and its DPP readers) */ \
    "v_add_u32_dpp v105, v102, v102 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v106, v102, %[M1] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v107, v102, %[M2] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp v105, v102, v105 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n" \
    "v_add3_u32 v109, v106, v107, v102\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_sub_co_u32_sdwa v110, vcc, %[PH], v105 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"   /* phase - extra bits of the quad (byte 2 of the sum); vcc = they reach below the cursor word */ \
    "v_alignbit_b32 " WIN ", %[W2], v125, %[PH]\n" \
    "v_cndmask_b32 v111, %[W2], v125, vcc\n" \
    "v_cndmask_b32 v112, v125, v124, vcc\n" \
    "v_alignbit_b32 v113, v111, v112, v110\n" \
    "v_bfe_u32 v116, v113, v109, v100\n" \
    "v_lshl_or_b32 %[S], v103, v100, v116\n" \
    "v_lshl_add_u32 v117, %[S], 1, %[TB]\n" \
    "ds_read_u16_d16_hi %[E], v117\n"
/* In the shadow of the table load (about 25 ns, a dozen instructions): the state word of the next record is
   formed, the cursor moves, the records of two steps go out in one 16-byte store, the ring words of the
   next step are requested.  Every lane stores: lanes 0..2 of a live quad the same records to the same
   address, all others into the sink. */
#ifdef CZC_EXP_NOSTORE   /* diagnostic builds (make exp): what the record store costs */
#define CZC_ASM_STORE(OFF) "s_nop 0\n"
#elif defined(CZC_EXP_SC1)   /* diagnostic: write-through record stores */
#define CZC_ASM_STORE(OFF) "global_store_dwordx4 %[RP], v[120:123], off offset:" #OFF " sc1\n"
#elif defined(CZC_EXP_SC01)
#define CZC_ASM_STORE(OFF) "global_store_dwordx4 %[RP], v[120:123], off offset:" #OFF " sc0 sc1\n"
#elif defined(CZC_EXP_NT)
#define CZC_ASM_STORE(OFF) "global_store_dwordx4 %[RP], v[120:123], off offset:" #OFF " nt\n"
#else
#define CZC_ASM_STORE(OFF) "global_store_dwordx4 %[RP], v[120:123], off offset:" #OFF "\n"
#endif
#define CZC_ASM_RING01 "ds_read2_b32 v[124:125], v115 offset1:1\n"
#define CZC_ASM_LGKM_E "s_waitcnt lgkmcnt(2)\n"
#define CZC_ASM_TAIL(STORE, NEXTH) \
    STORE \
    "v_lshl_add_u32 v118, %[S], %[SH], %[NK]\n" \
    "v_dot4c_i32_i8_e32 %[U], 0x01ff0001, v105\n" \
    "v_max_u32_sdwa %[SLOW], %[SLOW], v105 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"   /* most extra bits of any step so far; also the spacer: v118 -> DPP two instructions, v_dot4c -> reader three, 16-byte store -> write of its data one */ \
    "v_add_u32_dpp " NEXTH ", v118, v118 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp " NEXTH ", v118, " NEXTH " quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n" \
    "v_bfe_u32 v114, %[U], 5, 7\n" \
    "v_lshl_add_u32 v115, v114, 2, %[RB]\n" \
    CZC_ASM_RING01 \
    "ds_read_b32 %[W2], v115 offset:8\n"
#define CZC_ASM_EVEN(OFF) CZC_ASM_HEAD("v120") CZC_ASM_TAIL("", "v123")
#define CZC_ASM_ODD(OFF) CZC_ASM_HEAD("v122") CZC_ASM_TAIL(CZC_ASM_STORE(OFF), "v121")
#define CZC_ASM_PAIR(A) CZC_ASM_EVEN(A) CZC_ASM_ODD(A)
/* `rec`: where this lane's 32 records go (lanes 0..2 of a live quad: the chain's records; every other lane: the sink) */
__device__ static inline void czc_group_asm(CzcLane& c, const CzcRole& ro, CZ_GLOBAL uint64_t* rec) {
    static_assert(CZC_STEPS == 32, "the asm block is unrolled for 32 steps");
    /* ds_* instructions take the 32-bit LDS address */
    const uint32_t tb32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t*)ro.tb;
    const uint32_t rb32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)ro.ringm8;
    /* Inside the block the state register holds state + 512: (v << num_bits) | bits keeps the marker of v at
       bit 9, which the table base (tb32 - 1024) and the record word (nk = -(512 << sh)) absorb.
       Per lane p = extra bits << 16 | (64 - num_bits) [| 64 << 24 on lane 0]: summed over the quad, byte 0 is
       192 - state bits (its low 5 bits, summed over the lanes up to this one, are the v_bfe offset of this
       lane's field counted from the top of the window), byte 2 the extra bits, byte 3 the 64 that v_dot4c
       with weights (+1, 0, -1, +1) needs to turn byte 0 (-64 - state bits as a signed byte) into -state bits. */
    const uint32_t tbm = tb32 - 1024u, nk = 0u - (512u << ro.sh), k64 = ro.lane0 ? 0x40000040u : 0x40u;
    c.S += 512u;
    asm volatile(
        "v_mov_b32 v124, %[W0]\n"
        "v_mov_b32 v125, %[W1]\n"
        "v_mov_b32 v108, 0\n"
        /* state word of the first record */
        "v_lshl_add_u32 v118, %[S], %[SH], %[NK]\n"
        "s_nop 1\n"
        "v_add_u32_dpp v121, v118, v118 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
        "v_add_u32_dpp v121, v118, v121 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n"
        CZC_ASM_PAIR(0) CZC_ASM_PAIR(16) CZC_ASM_PAIR(32) CZC_ASM_PAIR(48) CZC_ASM_PAIR(64) CZC_ASM_PAIR(80) CZC_ASM_PAIR(96) CZC_ASM_PAIR(112)
        CZC_ASM_PAIR(128) CZC_ASM_PAIR(144) CZC_ASM_PAIR(160) CZC_ASM_PAIR(176) CZC_ASM_PAIR(192) CZC_ASM_PAIR(208) CZC_ASM_PAIR(224) CZC_ASM_PAIR(240)
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 %[PH], 31, %[U]\n"
        "v_mov_b32 %[W0], v124\n"
        "v_mov_b32 %[W1], v125\n"
        : [E] "+v"(c.E), [S] "+v"(c.S), [U] "+v"(c.u), [PH] "+v"(c.ph), [W0] "+v"(c.w0), [W1] "+v"(c.w1), [W2] "+v"(c.w2), [SLOW] "+v"(c.slow)
        : [TB] "v"(tbm), [RB] "v"(rb32), [M1] "v"(ro.m1 ? 0xFFu : 0u), [M2] "v"(ro.m2 ? 0xFFu : 0u), [SH] "v"(ro.sh), [NK] "v"(nk), [K64] "v"(k64),
          [XM] "s"(0x1F0000u), [RP] "v"(rec)
        : "memory", "vcc",
          "v100", "v101", "v102", "v103", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125");
    c.S -= 512u;
}

/* Round 5: the same 32 steps with the field picked by two 64-bit shifts instead of a compare, two selects, a funnel shift and a
 * bit-field extract — 29.5 instructions per step instead of 31.5, and a lone wave pays ~2.9 ns for every one of them
 * (profiles/r5/NOTES.md section 1; scripts/micro/chain_step.hip variants 1 / 10: 74.0 -> 67.2 ns).
 *   per lane p = num_bits << 8 | extra bits (v102); summed over the quad (v105): byte 0 = the sequence's extra bits, byte 1 = its state
 *   bits; sh (v109, low 6 bits) = extra bits + state bits of the lanes before this one = where this lane's field begins, counted
 *   from the cursor.  The 64 stream bits below the cursor {v128 = low, v129 = high} are shifted left by sh, the high word joins
 *   E >> 22 in {v130, v131} and that pair is shifted left by num_bits: v133 = (v << num_bits) | field = next state (+ 512, the marker
 *   of v).  64-bit operands must sit in even-aligned register pairs on gfx950: hence the two moves.  v_alignbit takes the low five
 *   bits of the cursor itself as its shift: no phase register.  A sequence of more than 32 extra bits leaves garbage in the state
 *   (still inside the table: v << num_bits | num_bits bits) and in the cursor; c.slow tells, and the group is redone wide.
 *   Inside the block the state lives in v133; the operand S is read once (first record) and written at the end. */
#define CZC_ASM2_HEAD(WIN) \
    CZC_ASM_LGKM_E \
    "v_ffbh_u32 v100, %[E]\n" \
    "v_bfe_u32 v101, %[E], 16, 5\n" \
    "v_lshl_or_b32 v102, v100, 8, v101\n" \
    "v_lshrrev_b32 v131, 22, %[E]\n" \
    "v_and_b32_dpp v106, v100, %[M1] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v107, v100, %[M2] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp v105, v102, v102 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp v105, v102, v105 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n" \
    "v_add3_u32 v109, v106, v107, v105\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_alignbit_b32 " WIN ", %[W2], v125, %[U]\n" \
    "v_alignbit_b32 v128, v125, v124, %[U]\n" \
    "v_mov_b32 v129, " WIN "\n" \
    "v_lshlrev_b64 v[128:129], v109, v[128:129]\n" \
    "v_mov_b32 v130, v129\n" \
    "v_lshlrev_b64 v[132:133], v100, v[130:131]\n" \
    "v_lshl_add_u32 v117, v133, 1, %[TB]\n" \
    "ds_read_u16_d16_hi %[E], v117\n"
#define CZC_ASM2_TAIL(STORE, NEXTH) \
    STORE \
    "v_lshl_add_u32 v118, v133, %[SH], %[NK]\n" \
    "v_dot4c_i32_i8_e32 %[U], 0xffff, v105\n" \
    "v_max_u32_sdwa %[SLOW], %[SLOW], v105 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"   /* most extra bits of any step so far; also the spacer: v118 -> DPP two instructions, v_dot4c -> reader three, 16-byte store -> write of its data one */ \
    "v_add_u32_dpp " NEXTH ", v118, v118 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp " NEXTH ", v118, " NEXTH " quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n" \
    "v_bfe_u32 v114, %[U], 5, 7\n" \
    "v_lshl_add_u32 v115, v114, 2, %[RB]\n" \
    CZC_ASM_RING01 \
    "ds_read_b32 %[W2], v115 offset:8\n"
#define CZC_ASM2_EVEN(OFF) CZC_ASM2_HEAD("v120") CZC_ASM2_TAIL("", "v123")
#define CZC_ASM2_ODD(OFF) CZC_ASM2_HEAD("v122") CZC_ASM2_TAIL(CZC_ASM_STORE(OFF), "v121")
#define CZC_ASM2_PAIR(A) CZC_ASM2_EVEN(A) CZC_ASM2_ODD(A)
__device__ static inline void czc_group_asm2(CzcLane& c, const CzcRole& ro, CZ_GLOBAL uint64_t* rec) {
    static_assert(CZC_STEPS == 32, "the asm block is unrolled for 32 steps");
    const uint32_t tb32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t*)ro.tb;
    const uint32_t rb32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)ro.ringm8;
    const uint32_t tbm = tb32 - 1024u, nk = 0u - (512u << ro.sh);
    c.S += 512u;
    asm volatile(
        "v_mov_b32 v124, %[W0]\n"
        "v_mov_b32 v125, %[W1]\n"
        "v_mov_b32 v133, %[S]\n"
        /* state word of the first record */
        "v_lshl_add_u32 v118, %[S], %[SH], %[NK]\n"
        "s_nop 1\n"
        "v_add_u32_dpp v121, v118, v118 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
        "v_add_u32_dpp v121, v118, v121 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n"
        CZC_ASM2_PAIR(0) CZC_ASM2_PAIR(16) CZC_ASM2_PAIR(32) CZC_ASM2_PAIR(48) CZC_ASM2_PAIR(64) CZC_ASM2_PAIR(80) CZC_ASM2_PAIR(96) CZC_ASM2_PAIR(112)
        CZC_ASM2_PAIR(128) CZC_ASM2_PAIR(144) CZC_ASM2_PAIR(160) CZC_ASM2_PAIR(176) CZC_ASM2_PAIR(192) CZC_ASM2_PAIR(208) CZC_ASM2_PAIR(224) CZC_ASM2_PAIR(240)
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 %[PH], 31, %[U]\n"
        "v_mov_b32 %[W0], v124\n"
        "v_mov_b32 %[W1], v125\n"
        "v_mov_b32 %[S], v133\n"
        : [E] "+v"(c.E), [S] "+v"(c.S), [U] "+v"(c.u), [PH] "+v"(c.ph), [W0] "+v"(c.w0), [W1] "+v"(c.w1), [W2] "+v"(c.w2), [SLOW] "+v"(c.slow)
        : [TB] "v"(tbm), [RB] "v"(rb32), [M1] "v"(ro.m1 ? 0xFFu : 0u), [M2] "v"(ro.m2 ? 0xFFu : 0u), [SH] "v"(ro.sh), [NK] "v"(nk), [RP] "v"(rec)
        : "memory",
          "v100", "v101", "v102", "v105", "v106", "v107", "v109", "v114", "v115", "v117", "v118", "v120", "v121", "v122", "v123", "v124", "v125",
          "v128", "v129", "v130", "v131", "v132", "v133");
    c.S -= 512u;
}

/* The same group for blocks that hold sequences of more than 32 extra bits (up to 16 + 16 + 31): FOUR ring words are kept
 * (v[124:127] = words -3 .. 0 at the cursor), the 32 bits behind the extra bits are picked from three word pairs instead of
 * two, and the record of a wide sequence carries its bit position (low word) and the CZC_REC_WIDE flag instead of the
 * stream bits.  About eight instructions per step more than the narrow group: a block switches to it at its first wide
 * sequence (the narrow group that met it is redone).  v119 position, v128 record flag, s[12:13] "this sequence is wide". */
#define CZC_ASMW_HEAD(WIN) \
    "s_waitcnt lgkmcnt(2)\n" \
    "v_ffbh_u32 v100, %[E]\n" \
    "v_and_or_b32 v101, %[E], %[XM], %[K64]\n" \
    "v_sub_u32 v102, v101, v100\n" \
    "v_lshrrev_b32 v103, 22, %[E]\n" \
    "v_sub_u32 v119, %[U], %[SB]\n"               /* bits still unread before this sequence */ \
    "v_add_u32_dpp v105, v102, v102 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v106, v102, %[M1] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n" \
    "v_and_b32_dpp v107, v102, %[M2] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp v105, v102, v105 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n" \
    "v_add3_u32 v109, v106, v107, v102\n" \
    "v_bfe_u32 v108, v105, 16, 8\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_cmp_ge_u32 vcc, %[PH], v108\n" \
    "v_sub_u32 v110, %[PH], v108\n" \
    "v_alignbit_b32 " WIN ", v127, v126, %[PH]\n" \
    "v_cndmask_b32 v111, v126, v127, vcc\n" \
    "v_cndmask_b32 v112, v125, v126, vcc\n" \
    "v_cmp_gt_i32 vcc, %[N32], v110\n"            /* the 32 bits start more than one word below the cursor word */ \
    "v_cmp_lt_u32_e64 s[12:13], 32, v108\n" \
    "s_nop 0\n" \
    "v_cndmask_b32 v111, v111, v125, vcc\n" \
    "v_cndmask_b32 v112, v112, v124, vcc\n" \
    "v_alignbit_b32 v113, v111, v112, v110\n" \
    "v_bfe_u32 v116, v113, v109, v100\n" \
    "v_lshl_or_b32 %[S], v103, v100, v116\n" \
    "v_lshl_add_u32 v117, %[S], 1, %[TB]\n" \
    "ds_read_u16_d16_hi %[E], v117\n"
#define CZC_ASMW_TAIL(STORE, WIN, H, NEXTH) \
    "v_cndmask_b32_e64 " WIN ", " WIN ", v119, s[12:13]\n" \
    "v_cndmask_b32_e64 v128, 0, %[FL], s[12:13]\n" \
    "v_or_b32 " H ", " H ", v128\n" \
    STORE \
    "v_lshl_add_u32 v118, %[S], %[SH], %[NK]\n" \
    "v_dot4c_i32_i8_e32 %[U], 0x01ff0001, v105\n" \
    "v_and_b32 v129, 31, v108\n" \
    "v_add_u32_dpp " NEXTH ", v118, v118 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n" \
    "v_add_u32_dpp " NEXTH ", v118, " NEXTH " quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n" \
    "v_bfe_u32 v114, %[U], 5, 7\n" \
    "v_and_b32 %[PH], 31, %[U]\n" \
    "v_lshl_add_u32 v115, v114, 2, %[RB]\n" \
    "ds_read2_b32 v[124:125], v115 offset1:1\n" \
    "ds_read2_b32 v[126:127], v115 offset0:2 offset1:3\n"
#define CZC_ASMW_EVEN(OFF) CZC_ASMW_HEAD("v120") CZC_ASMW_TAIL("", "v120", "v121", "v123")
#define CZC_ASMW_ODD(OFF) CZC_ASMW_HEAD("v122") CZC_ASMW_TAIL(CZC_ASM_STORE(OFF), "v122", "v123", "v121")
#define CZC_ASMW_PAIR(A) CZC_ASMW_EVEN(A) CZC_ASMW_ODD(A)
__device__ static inline void czc_group_asm_wide(CzcLane& c, const CzcRole& ro, CZ_GLOBAL uint64_t* rec) {
    const uint32_t tb32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t*)ro.tb;
    const uint32_t rb32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)ro.ringm8 - 4u;   /* ring - 12: four words end at the cursor word */
    const uint32_t tbm = tb32 - 1024u, nk = 0u - (512u << ro.sh), k64 = ro.lane0 ? 0x40000040u : 0x40u;
    c.S += 512u;
    asm volatile(
        "v_bfe_u32 v114, %[U], 5, 7\n"
        "v_lshl_add_u32 v115, v114, 2, %[RB]\n"
        "ds_read2_b32 v[124:125], v115 offset1:1\n"
        "ds_read2_b32 v[126:127], v115 offset0:2 offset1:3\n"
        /* state word of the first record */
        "v_lshl_add_u32 v118, %[S], %[SH], %[NK]\n"
        "s_nop 1\n"
        "v_add_u32_dpp v121, v118, v118 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf\n"
        "v_add_u32_dpp v121, v118, v121 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf\n"
        CZC_ASMW_PAIR(0) CZC_ASMW_PAIR(16) CZC_ASMW_PAIR(32) CZC_ASMW_PAIR(48) CZC_ASMW_PAIR(64) CZC_ASMW_PAIR(80) CZC_ASMW_PAIR(96) CZC_ASMW_PAIR(112)
        CZC_ASMW_PAIR(128) CZC_ASMW_PAIR(144) CZC_ASMW_PAIR(160) CZC_ASMW_PAIR(176) CZC_ASMW_PAIR(192) CZC_ASMW_PAIR(208) CZC_ASMW_PAIR(224) CZC_ASMW_PAIR(240)
        "s_waitcnt lgkmcnt(0)\n"
        : [E] "+v"(c.E), [S] "+v"(c.S), [U] "+v"(c.u), [PH] "+v"(c.ph)
        : [TB] "v"(tbm), [RB] "v"(rb32), [M1] "v"(ro.m1 ? 0xFFu : 0u), [M2] "v"(ro.m2 ? 0xFFu : 0u), [SH] "v"(ro.sh), [NK] "v"(nk), [K64] "v"(k64),
          [SB] "v"(ro.sbits), [FL] "v"(CZC_REC_WIDE), [XM] "s"(0x1F0000u), [N32] "s"(-32), [RP] "v"(rec)
        : "memory", "vcc", "s12", "s13",
          "v100", "v101", "v102", "v103", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129");
    c.S -= 512u;
    czc_ring_words(c, ro);                                              /* the three words the narrow group and its callers keep */
}
#endif

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
/* value held by lane 0 of this lane's quad (every lane takes part) */
__device__ static inline uint32_t czc_q0(uint32_t v) { return (uint32_t)__shfl((int)v, LANE & ~3); }
__device__ static inline uint64_t czc_q0_64(uint64_t v) { return ((uint64_t)czc_q0((uint32_t)(v >> 32)) << 32) | czc_q0((uint32_t)v); }

/* ---- cz_scan_kernel: the work lists of the pre-pass -------------------------------------------------------------------
 * The blocks of a frame are chained for EXECUTION (window, offset history), but three parts of the work need none of that:
 *   - the FSE chain of a block needs that block's bitstream and its three tables — described in the block itself or, in Repeat
 *     mode, in an earlier block of the frame (cz_chain_kernel, unit = block);
 *   - Huffman-coded literals need the block's streams and a tree — its own or, Treeless, that of an earlier block
 *     (literals_section_decoder.cairo:58-117; cz_huf_kernel, unit = block);
 *   - Raw and RLE blocks, and the literals of blocks without sequences, ahead of the frame's first block WITH sequences land
 *     at a place that follows from the headers alone (block_decoder.cairo:95-122, :229-232; cz_tile_kernel and, for Huffman
 *     literals, cz_huf_kernel writing straight into the output).
 * One lane per frame walks the frame's headers (no decoding: block header, literals-section header, sequences header:
 * block_decoder.cairo:237-321, literals_section.cairo:81-175, sequence_section.cairo:77-114).  Two passes over the same walk:
 *   pass 0  counts, per size class, the blocks with sequences and the Huffman sections, the copy runs, and the arena space
 *           (records, literal nodes) every frame needs
 *   pass 1  places each list entry — descending size class: entries of similar length end up in the same waves / the
 *           longest work starts first — allocates the frame's records and literal nodes, links them in frame order and sets
 *           frame_first[f], lit_first[f], frame_pre[f]
 * A frame is listed only if it is regular to its last block (no malformed header, truncated block, Repeat / Treeless of
 * something nothing defined, output beyond the caller's capacity, first sequences section shorter than chain_min_nseq ...)
 * and its entries fit the lists and arenas: otherwise frame_first[f] = lit_first[f] = frame_pre[f] = 0, its entries are void
 * and cz_decode_frames_kernel does the frame by itself.  Nothing here reports errors. */
struct CzsBlk { uint32_t type, blk_off, bsize;                          /* 0 Raw, 1 RLE, 2 Compressed; content offset in the frame, Block_Size */
                uint32_t lt, regen, lit_hdr, nseq, sbody, modes; };     /* Compressed: literals type, regenerated size, header bytes, sequences, first table description, modes */
/* the walk of one lane over its frame; cz_scan_kernel advances all lanes of a wave together, one block at a time,
   so that the counters they share are bumped once per wave and size class (and the CPU emulator sees uniform control flow) */
struct CzsWalk { const uint8_t* src; uint64_t len, pos, end, h8; uint32_t h8sh, defined, has_checksum; int first, active, ok, have_tree; };   /* h8 >> (8 * h8sh), or 0 when h8sh >= 8: the 8 bytes at pos, asked for a block ahead */
/* up to 8 bytes at `at` (little endian), zeros beyond the end of the frame (len >= 8).  ONE load whatever the position — the
   last 8 bytes of the frame shifted down when `at` is nearer to its end: a byte-by-byte tail path would be taken by some lane
   of the wave in nearly every step of the walk, and it costs every lane eight dependent loads. */
__device__ static inline uint64_t czs_ld8_issue(const uint8_t* src, uint64_t len, uint64_t at, uint32_t* sh) {   /* the load; its use (czs_ld8_value) may come much later */
    const uint64_t a2 = at + 8 <= len ? at : len - 8;
    uint64_t v; __builtin_memcpy(&v, src + a2, 8);
    *sh = (uint32_t)(at - a2);                                          /* 0 when the load is where it was asked for */
    return v;
}
__device__ static inline uint64_t czs_ld8_value(uint64_t raw, uint32_t sh) { return sh >= 8 ? 0 : raw >> (8 * sh); }
__device__ static inline uint64_t czs_ld8(const uint8_t* src, uint64_t len, uint64_t at) { uint32_t sh; const uint64_t v = czs_ld8_issue(src, len, at, &sh); return czs_ld8_value(v, sh); }
__device__ static inline void czs_begin(CzsWalk& w, const uint8_t* src, uint64_t len, int valid) {
    w.src = src; w.len = len; w.pos = 0; w.end = 0; w.h8 = 0; w.h8sh = 8; w.defined = 0; w.has_checksum = 0; w.first = 1; w.active = 0; w.ok = 0; w.have_tree = 0;
    if (!valid || len < 8 || len >= 0xFFFFFFF0ull) return;              /* block offsets are 32 bits; a frame with a block is 8 bytes at least (shorter ones: the decode kernel) */
    const uint64_t h0 = czs_ld8(src, len, 0);
    const uint32_t magic = (uint32_t)h0;
    const uint32_t d = (uint32_t)(h0 >> 32) & 0xFF;
    const uint32_t single = (d >> 5) & 1, didf = d & 3, dl = didf == 3 ? 4 : didf, flag = d >> 6;
    const uint32_t fl = flag == 0 ? (single ? 1u : 0u) : flag == 1 ? 2u : flag == 2 ? 4u : 8u;
    const uint32_t hl = 5 + (single ? 0 : 1) + dl + fl;
    if (magic != 0xFD2FB528u || len < hl) return;                       /* frame.cairo:152-284: only the header's length and validity matter here */
    if (!single) { const uint32_t wd = (uint32_t)(h0 >> 40) & 0xFF; const uint64_t base = 1ull << (10 + (wd >> 3)); if (base + (base / 8) * (wd & 7) >= 4123168604160ull) return; }
    w.pos = hl; w.active = 1; w.has_checksum = (d >> 2) & 1; w.h8 = czs_ld8_issue(src, len, hl, &w.h8sh);
}
/* advances to the next block: 1 with `b` filled, or 0 — the walk is over (w.active = 0) and w.ok says whether the frame was
   regular to its end */
__device__ static inline int czs_next(CzsWalk& w, uint32_t chain_min_nseq, CzsBlk& b) {
    const uint8_t* src = w.src; const uint64_t len = w.len;
    if (!w.active) return 0;
    do {                                                                /* block_decoder.cairo:237-321 */
        uint64_t pos = w.pos;
        if (pos == ~0ull) { w.active = 0; w.ok = 1; return 0; }          /* the last block was the frame's last */
        if (len - pos < 3) break;
        /* the block header and the literals section header behind it came with ONE load, issued while the previous block was
           looked at; the next block's are asked for as soon as this block's size is known, so the walk waits for memory once
           per block (the sequences header below is fetched beside it) */
        const uint64_t h = czs_ld8_value(w.h8, w.h8sh);
        const uint32_t b0 = (uint32_t)h & 0xFF, b1 = (uint32_t)(h >> 8) & 0xFF, b2 = (uint32_t)(h >> 16) & 0xFF;
        const uint32_t type = (b0 >> 1) & 3, size = (b0 >> 3) | (b1 << 5) | (b2 << 13), blast = b0 & 1;
        if (type == 3 || size > 128u * 1024u) break;
        const uint64_t body = pos + 3; const uint32_t content = type == 1 ? 1u : size;
        if (len - body < content) break;
        w.pos = blast ? ~0ull : body + content;
        if (blast) w.end = body + content;                              /* where the content checksum, if any, begins */
        else w.h8 = czs_ld8_issue(src, len, body + content, &w.h8sh);
        b.type = type; b.blk_off = (uint32_t)body; b.bsize = size; b.lt = 0; b.regen = 0; b.lit_hdr = 0; b.nseq = 0; b.sbody = 0; b.modes = 0;
        if (type != 2) return 1;
        /* literals section header (literals_section.cairo:81-175): sizes only */
        if (size == 0) break;
        const uint32_t l0 = (uint32_t)(h >> 24) & 0xFF, lt = l0 & 3, fmt = (l0 >> 2) & 3;
        const uint32_t need = lt <= 1 ? ((fmt == 0 || fmt == 2) ? 1u : (fmt == 1 ? 2u : 3u)) : (fmt <= 1 ? 3u : (fmt == 2 ? 4u : 5u));
        if (size < need) break;
        const uint32_t l1 = need > 1 ? (uint32_t)(h >> 32) & 0xFF : 0, l2 = need > 2 ? (uint32_t)(h >> 40) & 0xFF : 0, l3 = need > 3 ? (uint32_t)(h >> 48) & 0xFF : 0, l4 = need > 4 ? (uint32_t)(h >> 56) & 0xFF : 0;
        uint32_t upper, regen;
        if (lt <= 1) { regen = (fmt == 0 || fmt == 2) ? l0 >> 3 : (fmt == 1 ? (l0 >> 4) + (l1 << 4) : (l0 >> 4) + (l1 << 4) + (l2 << 12)); upper = lt == 1 ? 1u : regen; }
        else {
            regen = fmt <= 1 ? (l0 >> 4) + ((l1 & 0x3f) << 4) : (fmt == 2 ? (l0 >> 4) + (l1 << 4) + ((l2 & 3) << 12) : (l0 >> 4) + (l1 << 4) + ((l2 & 0x3f) << 12));
            upper = fmt <= 1 ? (l1 >> 6) + (l2 << 2) : (fmt == 2 ? (l2 >> 2) + (l3 << 6) : (l2 >> 6) + (l3 << 2) + (l4 << 10));
        }
        if (size - need < upper) break;
        if (lt == 3 && !w.have_tree) break;                             /* Treeless with no tree before it (literals_section_decoder.cairo:82-86) */
        if (lt == 2) w.have_tree = 1;
        const uint32_t so = need + upper, sl_ = size - so;              /* sequence_section.cairo:77-114 */
        if (sl_ == 0) break;
        const uint64_t sq = czs_ld8(src, len, body + so);               /* the sequences header: count (1..3 bytes) and the modes byte */
        const uint32_t s0 = (uint32_t)sq & 0xFF, s1 = (uint32_t)(sq >> 8) & 0xFF, s2 = (uint32_t)(sq >> 16) & 0xFF;
        uint32_t n = 0, hb = 0;
        if (s0 == 0) { }
        else if (s0 <= 127) { if (sl_ < 2) break; n = s0; hb = 1; }
        else if (s0 <= 254) { if (sl_ < 3) break; n = ((s0 - 128) << 8) + s1; hb = 2; }
        else { if (sl_ < 4) break; n = s1 + (s2 << 8) + 0x7F00u; hb = 3; }
        b.lt = lt; b.regen = regen; b.lit_hdr = need;
        if (!n) return 1;                                               /* (128, 0: no sequences but a modes byte) */
        if (w.first && n < chain_min_nseq) break;
        b.nseq = n; b.modes = (uint32_t)(sq >> (8 * hb)) & 0xFF; b.sbody = so + hb + 1;
        int undefined = 0;
        for (int t = 0; t < 3; t++) {                                   /* Repeat of a table nothing defined (sequence_section_decoder.cairo:483,551,643) */
            const uint32_t md = (b.modes >> (6 - 2 * t)) & 3;
            if (md == 3) { if (!((w.defined >> t) & 1u)) undefined = 1; } else w.defined |= 1u << t;
        }
        if (undefined) break;
        w.first = 0;
        return 1;
    } while (0);
    w.active = 0; w.ok = 0;                                             /* irregular: the frame is not pre-passed */
    return 0;
}
__device__ static inline uint32_t czs_class(uint32_t nseq) { return cz_hbs(nseq); }   /* 1..18 */
/* counters[cls] += 1 for every lane with `has`, one atomic per wave and class; returns the lane's ticket */
__device__ static inline uint32_t czs_ticket(uint32_t* counters, int has, uint32_t cls) {
    uint32_t ticket = 0;
    for (unsigned long long rem = __ballot(has); rem;) {
        const int leader = __ffsll((long long)rem) - 1;
        const uint32_t c0 = (uint32_t)__shfl((int)cls, leader);
        const unsigned long long m = __ballot(has && cls == c0);
        uint32_t base = 0;
        if (LANE == leader) base = atomicAdd(&counters[c0], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader);
        if (has && cls == c0) ticket = base + (uint32_t)__popcll(m & ((1ull << LANE) - 1ull));
        rem &= ~m;
    }
    return ticket;
}
/* The lists are filled by class (blocks by sequence count, literal sections by size, copies), and every lane of every wave adds
 * to them once per block of its frame.  One global atomic per block and class was the whole cost of the scan on batches of
 * many-block frames (230 000 atomics on 40 addresses: 0.3 ms per pass on the corpus-like mix): the wave counts in LDS instead
 * — czs_cnt[0..31] blocks per sequence-count class, [32..63] literal sections per size class, [64] copy runs — adds its totals
 * to the global counters once at the end of pass 0 and leaves them in scan_wave; pass 1 takes the wave's share of every class
 * range with one atomic at its start (czs_base) and hands out places from LDS.  Both passes walk the same blocks and make the
 * same decisions, so the counts agree. */
#define CZS_WAVE_WORDS 72u
__shared__ uint32_t czs_cnt[CZS_WAVE_WORDS], czs_base[CZS_WAVE_WORDS];
/* what one block contributes to the lists, the same in both passes.  `known`: the block's place in the output follows from the
   headers (no block with sequences before it); kout: that place */
struct CzsPlan { int chain, lit, copy; uint32_t copy_len, copy_fill; uint64_t copy_src; int direct; };
__device__ static inline CzsPlan czs_plan(const CzsBlk& b, int known) {
    CzsPlan q; q.chain = 0; q.lit = 0; q.copy = 0; q.copy_len = 0; q.copy_fill = 0; q.copy_src = 0; q.direct = 0;
    if (b.type != 2) { if (known && b.bsize) { q.copy = 1; q.copy_len = b.bsize; q.copy_fill = b.type == 1; q.copy_src = b.blk_off; } return q; }
    q.chain = b.nseq != 0;
    if (b.lt >= 2) { q.lit = 1; q.direct = known && !b.nseq; }
    else if (known && !b.nseq && b.regen) { q.copy = 1; q.copy_len = b.regen; q.copy_fill = b.lt == 1; q.copy_src = (uint64_t)b.blk_off + b.lit_hdr; }
    return q;
}
/* 64-bit inclusive prefix sum over the lanes and the wave total */
__device__ static inline uint64_t czs_scan64(uint64_t v, uint64_t* total) {
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t lo = __shfl_up((uint32_t)incl, (unsigned)d), hi = __shfl_up((uint32_t)(incl >> 32), (unsigned)d); if (LANE >= d) incl += ((uint64_t)hi << 32) | lo; }
    *total = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(incl >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)incl, 63);
    return incl;
}

extern "C" __global__ void __launch_bounds__(CZ_WG_THREADS) cz_scan_kernel(cz_batch_args a) {
    const uint32_t f = blockIdx.x * CZ_WG_THREADS + (uint32_t)LANE;
    const int valid = f < a.n;
    CzsWalk w;
    czs_begin(w, valid ? a.in_base + a.in_off[f] : nullptr, valid ? a.in_len[f] : 0, valid);
    CzsBlk b; b.type = b.blk_off = b.bsize = b.lt = b.regen = b.lit_hdr = b.nseq = b.sbody = b.modes = 0;
    const uint64_t ocap = valid ? a.out_cap[f] : 0;
    const int with_lits = a.lit_arena != nullptr;                       /* literal and copy lists only with a literal arena (cz_context_set_literal_arena) */
    /* the order in which the decode kernels take the frames: by size class of the compressed frame, largest first (one wave
       per frame there: the longest frames must not start last) */
    const uint32_t fcls = valid ? cz_hbs((uint32_t)(a.in_len[f] > 0xFFFFFFFFull ? 0xFFFFFFFFull : a.in_len[f])) : 0u;   /* 0..32 -> 0..31 */
    const uint32_t ft = czs_ticket(a.scan_ctl + (a.scan_pass == 0 ? 72 : 104), valid, fcls > 31 ? 31u : fcls);
    if (a.scan_pass == 1 && valid) {
        uint32_t fb = 0; for (uint32_t cc = 31; cc > (fcls > 31 ? 31u : fcls); cc--) fb += a.scan_ctl[72 + cc];
        a.frame_order[fb + ft] = f;
    }
    int known = 1; uint64_t kout = 0; uint32_t pre_blocks = 0;
    for (uint32_t i = (uint32_t)LANE; i < CZS_WAVE_WORDS; i += 64u) {
        czs_cnt[i] = 0; czs_base[i] = 0;
        if (a.scan_pass == 1) {                                         /* the wave's share of every class range: what it counted in pass 0, taken with one atomic */
            const uint32_t v = a.scan_wave[(uint64_t)blockIdx.x * CZS_WAVE_WORDS + i];
            if (v) czs_base[i] = atomicAdd(&a.scan_ctl[i < 32u ? 32u + i : (i < 64u ? 168u + (i - 32u) : 202u)], v);
        }
    }
    __syncthreads();
    if (a.scan_pass == 0) {
        uint64_t units = 0, lbytes = 0; int seen_seq = 0;
        while (__ballot(w.active)) {
            const int has = czs_next(w, a.chain_min_nseq, b);
            CzsPlan q = czs_plan(b, known);
            if (!has) { q.chain = q.lit = q.copy = 0; }
            if (has) {
                const uint64_t out = b.type != 2 ? b.bsize : (b.nseq ? 0 : b.regen);
                if (known && kout + out > ocap) { w.active = 0; w.ok = 0; q.chain = q.lit = q.copy = 0; }   /* the decode kernel reports it */
                else if (b.type == 2 && b.nseq) known = 0; else if (known) kout += out;
            }
            if (q.chain) atomicAdd(&czs_cnt[czs_class(b.nseq)], 1u);   /* the wave's counts per class, in LDS (see czs_cnt) */
            if (with_lits) {
                if (q.lit) atomicAdd(&czs_cnt[32u + czs_class(b.regen)], 1u);
                if (q.copy) atomicAdd(&czs_cnt[64], 1u);
            }
            if (q.chain) { units += 4ull + CZC_MAP_WORDS + b.nseq; seen_seq = 1; }
            if (q.lit && !q.direct) lbytes += 16ull + ((b.regen + 15u) & ~15u);
        }
        __syncthreads();
        for (uint32_t i = (uint32_t)LANE; i < CZS_WAVE_WORDS; i += 64u) {   /* one global atomic per class the wave met, and the wave's counts for pass 1 */
            const uint32_t v = czs_cnt[i];
            a.scan_wave[(uint64_t)blockIdx.x * CZS_WAVE_WORDS + i] = v;
            if (v) atomicAdd(&a.scan_ctl[i < 32u ? i : (i < 64u ? 136u + (i - 32u) : 201u)], v);
        }
        if (valid) {                                                    /* between the passes: arena units / literal bytes the frame needs */
            a.frame_first[f] = w.ok ? units : 0;
            if (with_lits) { a.lit_first[f] = w.ok ? lbytes : 0; a.frame_pre[f] = w.ok ? CZ_PRE_REGULAR : 0; }
            (void)seen_seq;
        }
        return;
    }
    /* pass 1: the frames of the wave get their share of the arenas with one atomic each */
    const int ok0 = valid && (with_lits ? a.frame_pre[f] != 0 : a.frame_first[f] != 0);
    const uint64_t units = ok0 ? a.frame_first[f] : 0, lbytes = ok0 && with_lits ? a.lit_first[f] : 0;
    uint64_t at = 0, lat = 0; int placed = ok0;
    {
        uint64_t total, incl = czs_scan64(units, &total), wbase = 0;
        if (LANE == 0 && total) wbase = atomicAdd(a.chain_top, (unsigned long long)total);
        wbase = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(wbase >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)wbase, 0);
        if (units) { at = 64ull + wbase + (incl - units); if (at + units > a.chain_capacity) { at = 0; placed = 0; } }   /* indices 0..63 are reserved: 0 = none, 8..39 = the sink of czc_group_asm */
    }
    if (with_lits) {
        uint64_t total, incl = czs_scan64(lbytes, &total), wbase = 0;
        if (LANE == 0 && total) wbase = atomicAdd(a.lit_top, (unsigned long long)total);
        wbase = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(wbase >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)wbase, 0);
        if (lbytes) { lat = 64ull + wbase + (incl - lbytes); if (lat + lbytes > a.lit_capacity) { lat = 0; placed = 0; } }   /* nodes start at offset 64 (lit_top counts from 0, like every word of the control block, which ONE memset clears per launch): offset 0 = "no node" */
    }
    if (!placed) { at = 0; lat = 0; }
    uint32_t base[20]; { uint32_t run = 0; for (int c = 19; c >= 0; c--) { base[c] = run; run += a.scan_ctl[c]; } }   /* larger classes first */
    uint32_t lbase[20]; { uint32_t run = 0; for (int c = 19; c >= 0; c--) { lbase[c] = run; run += with_lits ? a.scan_ctl[136 + c] : 0u; } }
    uint64_t first_hdr = 0, prev_hdr = 0, first_node = 0, prev_node = 0;
    int has_large = 0;                                                  /* a block of CZ_BIG_BLOCK_SEQS sequences or more: its chain runs in the launch of the large blocks */
    uint32_t defidx[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, tree_seg = 0xFFFFFFFFu;
    const uint64_t ibase = valid ? a.in_off[f] : 0, obase = valid ? a.out_off[f] : 0;
    while (__ballot(w.active)) {
        const int has = czs_next(w, a.chain_min_nseq, b);
        CzsPlan q = czs_plan(b, known);
        if (!has) { q.chain = q.lit = q.copy = 0; }
        const uint64_t kout_here = kout;
        if (has) {
            const uint64_t out = b.type != 2 ? b.bsize : (b.nseq ? 0 : b.regen);
            if (known && kout + out > ocap) { w.active = 0; w.ok = 0; q.chain = q.lit = q.copy = 0; }
            else if (b.type == 2 && b.nseq) known = 0; else if (known) { kout += out; pre_blocks++; }
        }
        /* every lane takes a place for everything it counted in pass 0, placed in the arenas or not: the wave's shares depend on it */
        const uint32_t cls = q.chain ? czs_class(b.nseq) : 0u;
        const uint32_t ticket = q.chain ? czs_base[cls] + atomicAdd(&czs_cnt[cls], 1u) : 0u;
        uint32_t lticket = 0, cticket = 0; const uint32_t lcls = q.lit ? czs_class(b.regen) : 0u;
        if (with_lits) {
            if (q.lit) lticket = czs_base[32u + lcls] + atomicAdd(&czs_cnt[32u + lcls], 1u);
            if (q.copy) cticket = czs_base[64] + atomicAdd(&czs_cnt[64], 1u);
        }
        if (q.chain) {
            uint32_t bb = 0; for (int c = 0; c < 20; c++) if ((uint32_t)c == cls) bb = base[c];
            const uint32_t idx = bb + ticket;
            if (idx >= a.blk_capacity) placed = 0;
            else {
                cz_blk_desc d; d.frame = f; d.blk_off = b.blk_off; d.bsize = b.bsize; d.nseq = at ? b.nseq : 0u; d.sbody = b.sbody; d.modes = b.modes; d.pad = 0; d.hdr = at;
                for (int t = 0; t < 3; t++) {
                    const uint32_t md = (b.modes >> (6 - 2 * t)) & 3;
                    d.def[t] = md == 3 ? defidx[t] : 0xFFFFFFFFu;
                    if (md != 3) defidx[t] = idx;
                }
                a.blk_desc[idx] = d;
                if (cls >= CZ_BIG_BLOCK_CLASS) has_large = 1;
                if (at) {
                    a.chain_arena[at + 2] = 0; a.chain_arena[at + 3] = 0;   /* word 3: "chain done" (cz_chain_kernel, the large blocks' launch) */
                    if (prev_hdr) a.chain_arena[prev_hdr + 2] = at; else first_hdr = at;
                    prev_hdr = at; at += 4ull + CZC_MAP_WORDS + b.nseq;
                }
            }
        }
        if (with_lits && q.lit) {
            uint32_t bb = 0; for (int c = 0; c < 20; c++) if ((uint32_t)c == lcls) bb = lbase[c];
            const uint32_t idx = bb + lticket;
            if (idx >= a.lit_seg_capacity) placed = 0;
            else {
                cz_lit_seg d; d.frame = placed ? f : 0xFFFFFFFFu; d.blk_off = b.blk_off; d.bsize = b.bsize; d.regen = b.regen; d.direct = (uint32_t)q.direct;
                d.def = b.lt == 3 ? tree_seg : 0xFFFFFFFFu;
                if (b.lt == 2) tree_seg = idx;
                d.dst = 0;
                if (q.direct) d.dst = obase + kout_here;
                else if (lat) {
                    /* the node: {next, regen} then the bytes, laid out and linked here so that cz_huf_kernel only fills it */
                    CZ_GLOBAL uint64_t* node = (CZ_GLOBAL uint64_t*)(a.lit_arena + lat);
                    node[0] = 0; node[1] = b.regen;
                    if (prev_node) *(CZ_GLOBAL uint64_t*)(a.lit_arena + prev_node) = lat; else first_node = lat;
                    prev_node = lat; d.dst = lat + 16; lat += 16ull + ((b.regen + 15u) & ~15u);
                } else d.frame = 0xFFFFFFFFu;
                a.lit_segs[idx] = d;
            }
        }
        if (with_lits && q.copy) {
            if (cticket >= a.copy_seg_capacity) placed = 0;
            else { cz_copy_seg d; d.src = ibase + q.copy_src; d.dst = obase + kout_here; d.len = placed ? q.copy_len : 0u; d.fill = q.copy_fill; a.copy_segs[cticket] = d; }
        }
    }
    int wx = 0, early = 0;
    if (valid) {
        const int good = placed && w.ok && pre_blocks <= CZ_PRE_COUNT;   /* (the count shares frame_pre[f] with the marks) */
        a.frame_first[f] = good ? first_hdr : 0;
        if (with_lits) {
            uint32_t pre = good ? (CZ_PRE_REGULAR | pre_blocks) : 0;
            /* a frame whose every block is done by the pre-pass kernels needs no walk by the decode kernels: its result record is
               written here (frame_decoder.cairo:189-200: finished, the stored checksum read) — unless its content checksum is to be
               verified, or the checksum is cut short (the decode kernel reports that).  cz_huf_kernel may still take the frame
               back (frame_pre[f] = 0): then cz_decode_frames_kernel does it from scratch and writes the record again. */
            if (good && known && !a.verify_checksum && (!w.has_checksum || w.len - w.end >= 4)) {
                pre |= CZ_PRE_DONE;
                uint32_t ck = 0, fl = CZ_RESULT_FINISHED; uint64_t pos = w.end;
                if (w.has_checksum) { ck = (uint32_t)czs_ld8(w.src, w.len, pos); fl |= CZ_RESULT_HAS_CHECKSUM; pos += 4; }
                cz_frame_result r; r.status = 0; r.blocks_decoded = pre_blocks; r.bytes_consumed = pos; r.bytes_produced = kout; r.checksum_from_data = ck; r.flags = fl;
                r.detail[0] = pre_blocks; r.detail[1] = pos; r.calculated_checksum = 0; r.reserved = 0;
                a.results[f] = r;
            }
            wx = good && !(pre & CZ_PRE_DONE) && first_hdr != 0 && units >= CZ_WX_MIN_UNITS && ocap < 0x80000000ull && !(a.verify_checksum && w.has_checksum) && a.wx_list != nullptr;
            int big = wx && units >= CZ_WX_BIG_UNITS;
            if (big) big = atomicAdd(&a.scan_ctl[210], 1u) < CZ_WX_BIG_MAX;
            /* a frame without a large block, and not one of the batch's large frames: the early execute launch may take it */
            early = good && !(pre & CZ_PRE_DONE) && first_hdr != 0 && !has_large && !big && a.exec_counter != nullptr;
            a.lit_first[f] = good ? (first_node ? first_node : 1) : 0; a.frame_pre[f] = pre | (wx ? CZ_PRE_WXLIST : 0u) | (big ? CZ_PRE_WXBIG : 0u) | (early ? CZ_PRE_EARLY : 0u);
        }
    }
    { const unsigned long long em = __ballot(early); if (em && LANE == __ffsll((long long)em) - 1) atomicAdd(&a.scan_ctl[211], (uint32_t)__popcll(em)); }
    if (a.wx_list) {                                                    /* frames for cz_wexec_kernel: everything pre-passed, output fits its LDS window, enough sequences for a workgroup */
        const unsigned long long wm = __ballot(wx);
        if (wm) {
            uint32_t base = 0;
            if (LANE == __ffsll((long long)wm) - 1) base = atomicAdd(&a.scan_ctl[206], (uint32_t)__popcll(wm));
            base = (uint32_t)__shfl((int)base, __ffsll((long long)wm) - 1);
            if (wx) a.wx_list[base + (uint32_t)__popcll(wm & ((1ull << LANE) - 1ull))] = f;
        }
    }
    if (with_lits) {                                                    /* frames the decode kernels still have to walk */
        const unsigned long long todo = __ballot(valid && !(a.frame_pre[f < a.n ? f : 0] & CZ_PRE_DONE));
        if (LANE == 0 && todo) atomicAdd(&a.scan_ctl[204], (uint32_t)__popcll(todo));
    }
}

/* ---- cz_chain_kernel ---------------------------------------------------------------------------------
 * Table descriptions of one block (sequence_section_decoder.cairo:405-647), serial, by the slot's owner lane.
 * want = tables to read (bit t: LL, OF, ML).  For a table in Predefined / FSE mode the normalised counts go to
 * probs[t] and info gets (symbols - 1) | log << 6 at bits 10 t; for RLE mode rle[t] = the symbol; Repeat leaves all
 * as they are.  stage: LDS copy of the content from offset sbody (stage_n bytes), or nullptr.  Returns 0 / 1 (irregular). */
__device__ static inline __attribute__((always_inline)) int czc_parse_tables(cz_gcptr blk, uint32_t bsize, uint32_t sbody, uint32_t modes, uint32_t want, const uint8_t* stage, uint32_t stage_n,
                                       int16_t (*probs)[CZC_MAXSYM], uint32_t* info, int32_t* rle, uint32_t* bitoff) {
    uint32_t off = sbody;
    for (int t = 0; t < 3; t++) {                                       /* LL, OF, ML */
        const uint32_t md = (modes >> (6 - 2 * t)) & 3, max_log = t == 1 ? 8u : 9u;
        const int take = (want >> t) & 1u;
        if (md == 0) {
            if (take) {
                const int8_t* d = t == 0 ? CZ_LL_DEFAULT : t == 1 ? CZ_OF_DEFAULT : CZ_ML_DEFAULT;
                const uint32_t n = t == 0 ? 36u : t == 1 ? 29u : 53u, lg = t == 1 ? 5u : 6u;
                for (uint32_t s = 0; s < n; s++) probs[t][s] = d[s];
                *info = (*info & ~(0x3FFu << (10 * t))) | (((n - 1) | (lg << 6)) << (10 * t));
                rle[t] = -1;
            }
        } else if (md == 1) {
            if (off >= bsize) return 1;
            const uint32_t sym = blk[off]; off += 1;
            if (take) {
                if (sym >= (t == 0 ? 36u : (t == 1 ? 32u : 53u))) return 1;
                rle[t] = (int32_t)sym; *info &= ~(0x3FFu << (10 * t));
            }
        } else if (md == 2) {
            CzFBits br; br.g = (cz_gcptr)(blk + off); br.len = bsize - off; br.idx = 0; br.stage = stage; br.stage_lo = 0; br.stage_hi = 0;
            if (stage && off - sbody < stage_n) { br.stage = stage + (off - sbody); br.stage_hi = stage_n - (off - sbody); }
            uint32_t np, lg, used;
            /* a description that is only stepped over is parsed into the counts of the (later) wanted table, which its own parse overwrites */
            const int into = take ? t : (want & 2u ? 1 : 2);
            if (cz_fse_read_probs_inl(br, max_log, probs[into], &np, &lg, &used, 100, CZC_MAXSYM) || np > CZC_MAXSYM || np == 0) return 1;
            if (take) { *info = (*info & ~(0x3FFu << (10 * t))) | (((np - 1) | (lg << 6)) << (10 * t)); rle[t] = -1; }
            off += used;
            if (off > bsize) return 1;
        }
        if ((want >> (t + 1)) == 0) break;                              /* nothing behind this table is wanted */
    }
    *bitoff = off;
    return 0;
}

/* The workgroup of cz_chain_kernel is ONE wave, and the LDS executes a wave's accesses in order: between the phases that hand LDS
 * contents from some lanes to others (stage -> parse -> build, ring top-up -> ring words) nothing has to be waited for, the compiler
 * only must not move the accesses.  __syncthreads() also waits for every global access in flight (its fence is s_waitcnt vmcnt(0)):
 * at a ring top-up — one per four groups on config 4a — that was the wave's record stores of the last microsecond, 8.6 % of the
 * kernel's wave time (profiles/r5/NOTES.md).  cz_wave_sync: wavefront-scope fences and a wave barrier (the CPU emulator: a real
 * barrier of the wave's 64 threads). */
#define CZC_SYNC() cz_wave_sync()
extern "C" __global__ void __launch_bounds__(CZ_WG_THREADS, 1) cz_chain_kernel(cz_batch_args a) {
    __shared__ CzChainShared cs;
    __builtin_amdgcn_s_setprio(3);                                      /* the launch lasts as long as its longest chain: its waves issue ahead of cz_huf1_kernel's beside them */
    for (uint32_t i = (uint32_t)LANE; i < 36; i += 64) cs.llml[i] = CZ_LL_BASE[i] | ((uint32_t)CZ_LL_BITS[i] << 24);
    for (uint32_t i = (uint32_t)LANE; i < 53; i += 64) cs.llml[40 + i] = CZ_ML_BASE[i] | ((uint32_t)CZ_ML_BITS[i] << 24);
    if (LANE < 2) cs.idle[LANE] = (uint16_t)CZC_E16_IDLE;
    __syncthreads();
    const uint32_t qk = (uint32_t)LANE >> 2, ql = (uint32_t)LANE & 3u;  /* slot, lane of the quad: 0 LL (owner), 1 ML, 2 OF, 3 idle */
    const int has_slot = qk < CZC_SLOTS;
    const int owner = has_slot && ql == 0;                              /* does the serial parsing of its slot's block */
    CzChainSlot& sl = cs.slot[has_slot ? qk : 0];
    CzcRole ro;
    ro.ringm8 = sl.ring - 8; ro.tb = cs.idle; ro.sbits = 0;
    ro.m1 = (ql == 1 || ql == 2) ? 0xFF00u : 0u; ro.m2 = ql == 2 ? 0xFF00u : 0u; ro.sh = ql == 3 ? 0u : 9u * ql; ro.lane0 = ql == 0;
    const uint16_t* my_table = ql == 0 ? sl.t_ll : (ql == 1 ? sl.t_ml : sl.t_of);
#ifdef CZ_PROFILE
    unsigned long long cprof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ct_ = __builtin_amdgcn_s_memtime();
#endif
    uint32_t ndesc = 0, nlarge = 0; for (int c = 0; c < 20; c++) { ndesc += a.scan_ctl[c]; if ((uint32_t)c >= CZ_BIG_BLOCK_CLASS) nlarge += a.scan_ctl[c]; }
    if (ndesc > a.blk_capacity) ndesc = a.blk_capacity;
    if (nlarge > ndesc) nlarge = ndesc;
    /* this launch's share of the block list (largest class first, so the large blocks are its head) and its work counter;
       the launch of the large blocks publishes every block it is done with (CZ_RELEASE_AGENT, header word 3) */
    const uint32_t list_lo = a.chain_part == 2u ? nlarge : 0u, list_hi = a.chain_part == 1u ? nlarge : ndesc;
    uint32_t* const list_counter = a.chain_part == 2u ? &a.scan_ctl[212] : &a.scan_ctl[64];
    const int publish = a.chain_part == 1u;
    if (list_lo >= list_hi) { if (LANE == 0) atomicAdd(&a.scan_ctl[205], 1u); return; }   /* nothing for this launch: its waves leave at once */
    /* per slot (the same values in the four lanes of its quad, except where noted) */
    int chain_live = 0, wide = 0, exhausted = 0;                        /* exhausted: owner lanes */
    uint32_t qnseq = 0, done = 0, sbits = 0;
    uintptr_t S = 0, E = 0; intptr_t ring_base = 0, loaded_lo = 0;
    CZ_GLOBAL uint64_t* rec = nullptr;
    /* owner lanes: the block in hand */
    uint32_t o_frame = 0, o_nseq = 0, o_mapflags = 0, o_bitoff = 0, o_bad0 = 0; uint64_t o_hdr = 0; int o_have = 0, o_pub = 0;
    CzcLane c; c.E = (uint32_t)CZC_E16_IDLE << 16; c.S = 0; c.u = 0; c.ph = 0; c.w0 = c.w1 = c.w2 = 0; c.slow = 0;
    CzcPre pre;
    for (uint32_t r = 0; r < CZC_PF; r++) pre.v[r] = uint4{0, 0, 0, 0};
    for (;;) {
        /* ---- slots without a block take the next one of the list */
        if (__ballot(owner && !o_have && !exhausted)) {
            cz_gcptr blk = nullptr; uint32_t bsize = 0, sbody = 0, modes = 0; uint32_t def[3] = {0, 0, 0};
            int got = 0;
            if (owner && !o_have && !exhausted) {
                for (;;) {
                    const uint32_t idx = list_lo + atomicAdd(list_counter, 1u);
                    if (idx >= list_hi) { exhausted = 1; break; }
                    const cz_blk_desc d = a.blk_desc[idx];
                    if (!d.nseq) continue;                              /* void entry: its frame is not pre-passed */
                    blk = (cz_gcptr)(a.in_base + a.in_off[d.frame] + d.blk_off); bsize = d.bsize; sbody = d.sbody; modes = d.modes;
                    def[0] = d.def[0]; def[1] = d.def[1]; def[2] = d.def[2];
                    o_frame = d.frame; o_nseq = d.nseq; o_hdr = d.hdr; o_have = 1; got = 1;
                    o_pub = publish && (a.frame_pre[d.frame] & CZ_PRE_WXBIG) != 0u;   /* a block of one of the batch's large frames: cz_wexec_kernel's early launch may wait for it */
                    break;
                }
            }
            /* stage the head of the sequences section (table descriptions) linearly: 256 bytes */
            {
                const uintptr_t base = (uintptr_t)blk + sbody;
                const uintptr_t bk = (uintptr_t)czc_q0_64((uint64_t)base), Sk = (uintptr_t)czc_q0_64((uint64_t)(uintptr_t)blk), Ek = Sk + czc_q0(bsize);
                CZC_SYNC();                                        /* the slot's ring (same LDS) is no longer read */
                if (czc_q0((uint32_t)got) && has_slot) for (uint32_t cc = ql; cc < 16; cc += CZC_LPS) *(uint4*)&sl.stage[16 * cc] = czc_load16(bk + 16 * cc, Sk, Ek);
                CZC_SYNC();
            }
            CZC_PROF_ACC(8); CZC_PROF_CNT(13);
            /* tables (sequence_section_decoder.cairo:405-647), serial per owner lane; a Repeat mode reads the description of
               the block that defined the table (the tables in LDS belong to whatever block this slot had before) */
            uint32_t binfo = 0; int32_t rles[3] = {-1, -1, -1}; int bad = 0;
            if (got) {
                o_mapflags = 0;
                for (int t = 0; t < 3; t++) if (((modes >> (6 - 2 * t)) & 3) != 3) o_mapflags |= 1u << t;
                bad = czc_parse_tables(blk, bsize, sbody, modes, 7u, sl.stage, 256u, sl.probs, &binfo, rles, &o_bitoff);
            }
            CZC_PROF_ACC(9);
            /* Repeat modes: the description is in the block that defined the table; its head is staged like the block's own
               (three uniform rounds, one per table; byte loads from global memory would cost a round trip each) */
            for (int t = 0; t < 3; t++) {
                const int rep = got && !bad && ((modes >> (6 - 2 * t)) & 3) == 3;
                if (!__ballot(rep)) continue;
                cz_gcptr dblk = nullptr; uint32_t dbsize = 0, dsbody = 0, dmodes = 0;
                if (rep) {
                    if (def[t] >= ndesc) bad = 1;
                    else {
                        const cz_blk_desc dd = a.blk_desc[def[t]];
                        if (dd.frame != o_frame) bad = 1;
                        else { dblk = (cz_gcptr)(a.in_base + a.in_off[dd.frame] + dd.blk_off); dbsize = dd.bsize; dsbody = dd.sbody; dmodes = dd.modes; }
                    }
                }
                const int go = rep && !bad;
                {
                    const uintptr_t base = (uintptr_t)dblk + dsbody;
                    const uintptr_t bk = (uintptr_t)czc_q0_64((uint64_t)base), Sk = (uintptr_t)czc_q0_64((uint64_t)(uintptr_t)dblk), Ek = Sk + czc_q0(dbsize);
                    CZC_SYNC();
                    if (czc_q0((uint32_t)go) && has_slot) for (uint32_t cc = ql; cc < 16; cc += CZC_LPS) *(uint4*)&sl.stage[16 * cc] = czc_load16(bk + 16 * cc, Sk, Ek);
                    CZC_SYNC();
                }
                if (go) { uint32_t dummy; bad = czc_parse_tables(dblk, dbsize, dsbody, dmodes, 1u << t, sl.stage, 256u, sl.probs, &binfo, rles, &dummy); }
            }
            if (got && bad) { got = 0; o_have = 0; a.frame_first[o_frame] = 0; if (o_pub) CZ_ST_AGENT(&a.chain_arena[o_hdr + 3], (uint64_t)2); }
            /* What the batch's offsets look like, for the execute stage (cz_wx_side_by_side): sequences (in units of 64) weighted by
               the share of their block's offset codes that are near (2..13: offsets below 16 KiB, which a frame's waves in
               cz_wexec_kernel would have to wait on each other for) and far (14 and up). */
            if (got && a.exec_counter) {
                const uint32_t info = (binfo >> 10) & 0x3FFu, lg = info >> 6, np = (info & 63u) + 1u, n6 = (o_nseq + 63u) >> 6;
                uint32_t nearc = 0, farc = 0;
                if (rles[1] >= 0) { nearc = rles[1] >= 2 && rles[1] <= 13 ? 256u : 0u; farc = rles[1] >= 14 ? 256u : 0u; }
                else if (lg) {
                    for (uint32_t sy = 2; sy < np; sy++) { const int32_t pr = sl.probs[1][sy]; const uint32_t cnt = pr > 0 ? (uint32_t)pr : (pr < 0 ? 1u : 0u); if (sy <= 13u) nearc += cnt; else farc += cnt; }
                    nearc = (nearc << 8) >> lg; farc = (farc << 8) >> lg;
                }
                if (nearc) atomicAdd(a.chain_top + 5, (unsigned long long)nearc * n6);
                if (farc) atomicAdd(a.chain_top + 6, (unsigned long long)farc * n6);
                /* ... and the share of sequences the execute kernel's fast loop does not take: a literal run above 8 bytes (LL code > 8)
                   or a match above 16 (ML code > 13); the two shares are added (an upper bound) */
                uint32_t longc = 0;
                for (int t = 0; t < 3; t += 2) {
                    const uint32_t inf = (binfo >> (10 * t)) & 0x3FFu, lgt = inf >> 6, npt = (inf & 63u) + 1u, lim = t == 0 ? 8u : 13u;
                    if (rles[t] >= 0) longc += (uint32_t)rles[t] > lim ? 256u : 0u;
                    else if (lgt) {
                        uint32_t cl = 0;
                        for (uint32_t sy = lim + 1u; sy < npt; sy++) { const int32_t pr = sl.probs[t][sy]; cl += pr > 0 ? (uint32_t)pr : 0u; }   /* ("less than one" symbols: rare by definition, and encoders list symbols they never use that way) */
                        longc += (cl << 8) >> lgt;
                    }
                }
                if (longc) atomicAdd(a.chain_top + 7, (unsigned long long)longc * n6);
            }
            CZC_PROF_ACC(10);
            /* build the tables of the refilled slots, each by the whole wave, slot after slot: LL and ML first (the slot's OF
               table, always rewritten for a new block, is their scratch), then OF (scratch: the description bytes, now read) */
            {
                CZC_SYNC();                                        /* descriptions read */
                int tbad = 0;
                for (unsigned long long need = __ballot(owner && got); need; need &= need - 1) {
                    const int ol = cz_unii(__ffsll((long long)need) - 1);
                    const uint32_t binf = cz_readlane(binfo, ol), mf = cz_readlane(o_mapflags, ol);
                    const uint64_t hk = ((uint64_t)cz_readlane((uint32_t)(o_hdr >> 32), ol) << 32) | cz_readlane((uint32_t)o_hdr, ol);
                    CzChainSlot& bs = cs.slot[(uint32_t)ol >> 2];
                    cz_gptr maps = (cz_gptr)(a.chain_arena + hk + 4);
                    int sbad = 0;
#pragma unroll 1
                    for (uint32_t o = 0; o < 3; o++) {
                        const uint32_t bk = o == 0 ? 0u : (o == 1 ? 2u : 1u);   /* LL, ML, OF */
                        const uint32_t info = (binf >> (10 * bk)) & 0x3FFu;
                        if (!(info >> 6)) continue;
                        uint16_t* table = bk == 0 ? bs.t_ll : (bk == 1 ? bs.t_of : bs.t_ml);
                        uint8_t* symof = bk == 1 ? bs.stage : (uint8_t*)bs.t_of;
                        cz_gptr map = !((mf >> bk) & 1u) ? (cz_gptr)nullptr : (bk == 0 ? maps : (bk == 2 ? maps + 512 : maps + 1024));   /* Repeat: the decode kernel keeps the earlier map */
                        sbad |= czc_fse_build_wave(table, bs.probs[bk], (info & 63u) + 1u, info >> 6, symof, bs.counters_ml, cs.llml, bk, map);
                        cz_wave_sync();
                    }
                    if (sbad && LANE == ol) tbad = 1;
                }
                CZC_SYNC();
                if (got && tbad) { got = 0; o_have = 0; a.frame_first[o_frame] = 0; if (o_pub) CZ_ST_AGENT(&a.chain_arena[o_hdr + 3], (uint64_t)2); }
                if (got) for (int t = 0; t < 3; t++) if (rles[t] >= 0) {    /* RLE: a one-state table (num_bits 0, base 0) at state 0 (:437-446) */
                    uint16_t* table = t == 0 ? sl.t_ll : (t == 1 ? sl.t_of : sl.t_ml);
                    table[0] = CZC_E16(0u, 0u, t == 1 ? (uint32_t)rles[t] : (cs.llml[(t == 2 ? 40u : 0u) + (uint32_t)rles[t]] >> 24));
                    if ((o_mapflags >> t) & 1u) ((cz_gptr)(a.chain_arena + o_hdr + 4))[t == 0 ? 0 : (t == 2 ? 512 : 1024)] = (uint8_t)rles[t];
                }
                CZC_SYNC();
            }
            CZC_PROF_ACC(11);
            /* the bit ring of the new blocks: the top 256 bytes of the stream; from here on every lane of a quad holds its slot's values */
            const int qgot = (int)czc_q0((uint32_t)got) && has_slot;
            const uint32_t gn = czc_q0(o_nseq);
            const uintptr_t gblk = (uintptr_t)czc_q0_64((uint64_t)(uintptr_t)blk);
            const uint32_t gbitoff = czc_q0(o_bitoff), gbsize = czc_q0(bsize);
            if (qgot) {
                S = gblk + gbitoff; E = gblk + gbsize; qnseq = gn; done = 0; wide = 0;
                sbits = (uint32_t)(S & (CZC_RING - 1)) * 8u;
                ring_base = (intptr_t)(S & ~(uintptr_t)(CZC_RING - 1));   /* ring-space bit u <-> byte ring_base + (u >> 3) */
                const intptr_t hi = (intptr_t)((E + 15) & ~(uintptr_t)15);
                loaded_lo = hi - 256;
                czc_prefetch(pre, 1, hi, S, E);
                czc_commit(sl, 1, hi, loaded_lo, pre);
            }
            CZC_SYNC();
            if (qgot) czc_prefetch(pre, 1, loaded_lo, S, E);               /* the next 256 bytes: in registers long before they are needed */
            /* initial states (owner), handed to the lanes of the quad */
            int32_t u0 = 0; uint32_t st_ll = 0, st_of = 0, st_ml = 0;
            if (got) {
                int32_t p = (int32_t)(E - S) * 8; int skipped = 0;
                o_bad0 = 0;
                for (;;) {                                              /* padding :46-64 */
                    const uint32_t b = p > 0 ? (uint32_t)(czc_window(sl, (int32_t)sbits + p) >> 63) : 0; p -= 1; skipped++;
                    if (b == 1 || skipped > 8) break;
                }
                if (skipped > 8) o_bad0 = 1;
                uint32_t stv[3] = {0, 0, 0};
                for (int t = 0; t < 3; t++) {                           /* init order LL, OF, ML (:207-218) */
                    if (rles[t] >= 0) continue;
                    const uint32_t lg = (binfo >> (10 * t + 6)) & 15u;
                    stv[t] = p > 0 ? (uint32_t)(czc_window(sl, (int32_t)sbits + p) >> (64 - lg)) : 0; p -= (int32_t)lg;
                }
                st_ll = stv[0]; st_of = stv[1]; st_ml = stv[2];
                if (p < 0) o_bad0 = 1;
                u0 = (int32_t)sbits + p;
            }
            {
                const uint32_t qll = czc_q0(st_ll), qml = czc_q0(st_ml), qof = czc_q0(st_of), qu = czc_q0((uint32_t)u0);
                const uint64_t qh = czc_q0_64(o_hdr);
                if (qgot) {
                    c.u = (int32_t)qu; c.slow = 0; ro.sbits = sbits;
                    c.S = ql == 0 ? qll : (ql == 1 ? qml : (ql == 2 ? qof : 0u));
                    if (ql == 3) { ro.tb = cs.idle; c.S = 0; } else ro.tb = my_table;
                    c.E = (uint32_t)ro.tb[c.S] << 16;
                    czc_ring_words(c, ro);
                    rec = (CZ_GLOBAL uint64_t*)(a.chain_arena + qh + 4 + CZC_MAP_WORDS);
                    chain_live = 1;
                }
            }
            CZC_PROF_ACC(1);
        }
        if (!__ballot(chain_live)) { if (!__ballot(owner && !exhausted)) break; continue; }
        /* ---- keep CZC_NEED bytes below every live cursor staged; when one slot runs low, all top up */
        {
            const int32_t uq = (int32_t)czc_qp<CZC_QP(0, 0, 0, 0)>((uint32_t)c.u);   /* lane 3 of a quad does not follow the cursor */
            const intptr_t curb = ring_base + ((uq > 0 ? uq - 1 : 0) >> 3);    /* byte that holds the next unread bit */
            const int need = chain_live && (curb - (intptr_t)CZC_NEED < loaded_lo);
            if (__ballot(need)) {
                /* lowest start whose 512 bytes still cover the word at the cursor */
                intptr_t new_lo = chain_live ? ((curb - (intptr_t)(CZC_RING - 4) + 15) & ~(intptr_t)15) : loaded_lo;
                if (new_lo > loaded_lo) new_lo = loaded_lo;
                if (new_lo < loaded_lo - 256) new_lo = loaded_lo - 256;      /* czc_commit moves at most 16 pieces */
                CZC_SYNC();
                czc_commit(sl, chain_live, loaded_lo, new_lo, pre);
                loaded_lo = new_lo;
                CZC_SYNC();
                czc_prefetch(pre, chain_live, loaded_lo, S, E);
                if (chain_live) czc_ring_words(c, ro);                    /* the words under the cursor may just have arrived */
                CZC_PROF_ACC(2); CZC_PROF_CNT(5);
            }
        }
        /* ---- one group of steps */
        {
            /* uniform choice of the loop flavour: the scheduled asm group of CZC_STEPS steps unless a chain is near its end;
               a block that met a sequence of more than 32 extra bits goes on with the wide asm group (the narrow one is redone) */
            const uint32_t left = qnseq - done;
            const int tail = __ballot(chain_live && left <= CZC_STEPS) != 0;
            int ran_asm = 0;
            (void)tail; (void)wide;
#if defined(__HIP_DEVICE_COMPILE__)
            if (!tail && !(a.debug_flags & CZ_DEBUG_CHAIN_CPP_STEP)) {
                CZ_GLOBAL uint64_t* rp = chain_live && ql < 3 ? rec + done : (CZ_GLOBAL uint64_t*)(a.chain_arena + 8);
                if (!__ballot(chain_live && wide)) {
                    const CzcLane sv = c;
                    c.slow = 0;
#ifdef CZC_EXP_OLD_STEP
                    czc_group_asm(c, ro, rp);
#else
                    czc_group_asm2(c, ro, rp);
#endif
                    const int hit = chain_live && ql < 3 && c.slow > 32;
                    if (__ballot(hit)) { wide |= (int)czc_q0((uint32_t)hit); c = sv; }   /* redo this group wide; the slots that met a wide sequence stay wide for their block */
                }
                if (__ballot(chain_live && wide)) czc_group_asm_wide(c, ro, rp);
                if (chain_live) done += CZC_STEPS;
                ran_asm = 1;
            }
#endif
            if (!ran_asm) {
                const uint32_t steps = !chain_live ? 0u : (left < CZC_WIDE_STEPS ? left : CZC_WIDE_STEPS);
                for (uint32_t i = 0; i < CZC_WIDE_STEPS; i++) czc_step(c, ro, rec + done + i, i < steps, done + i + 1 == qnseq, ql == 0);
                done += steps;
            }
            CZC_PROF_ACC(3); CZC_PROF_CNT(6);
        }
        /* ---- finished chains: finalize the block (owner) and free the slot */
        if (__ballot(chain_live && done >= qnseq)) {
            const int fin = chain_live && done >= qnseq;
            int fbad = 0;
            if (fin && owner) {
                /* the cursor only moves down, so an overrun (NotEnoughBytes, :281) shows in its final value */
                fbad = o_bad0 || c.u - (int32_t)sbits != 0;
                if (fbad) a.frame_first[o_frame] = 0;                   /* padding / overrun / ExtraBits: the frame is not pre-passed */
                else { uint64_t* h = a.chain_arena + o_hdr; h[0] = ((uint64_t)o_nseq << 32) | o_mapflags; h[1] = o_bitoff; if (!o_pub) h[3] = 0; }
                o_have = 0;
            }
            if (__ballot(fin && owner && o_pub)) {
                /* the block's records, maps and header — stores of every lane of this wave — leave this XCD's L2 before the flag does:
                   cz_wexec_kernel's early launch, on another CU, starts on the block when it sees the flag.  (Only blocks of the batch's
                   LARGE frames — a few hundred at most — are published: the write-back is a whole L2's.) */
                CZ_RELEASE_AGENT();
                if (fin && owner && o_pub) CZ_ST_AGENT(&a.chain_arena[o_hdr + 3], (uint64_t)(fbad ? 2 : 1));
            }
            if (fin) { chain_live = 0; wide = 0; ro.tb = cs.idle; c.S = 0; c.E = (uint32_t)CZC_E16_IDLE << 16; }
            CZC_PROF_ACC(4);
        }
    }
#ifdef CZ_PROFILE
    if (LANE == 0 && a.prof) { for (int i = 0; i < 8; i++) atomicAdd(&a.prof[32 + i], cprof[i]); for (int i = 8; i < 14; i++) atomicAdd(&a.prof[50 + i - 8], cprof[i]); }
#endif
    if (LANE == 0) atomicAdd(&a.scan_ctl[205], 1u);                     /* counted out: cz_huf1_kernel leaves the rest of the literals to cz_huf_kernel once all waves have */
}
