/*
 * czstd_pre.hip — cz_huf_kernel + cz_tile_kernel: the parts of a frame that need neither its window nor its offset history,
 * done next to cz_chain_kernel from the lists cz_scan_kernel makes (czstd_chain.hip).
 *
 * cz_huf_kernel: Huffman-coded literals (literals_section_decoder.cairo:58-243, huff0_decoder.cairo:149-467), unit of work =
 * ONE literals section (a block), not a frame: a workgroup of CZH_WAVES waves takes the next list entry (largest regenerated
 * size first), wave 0 parses the section header and the tree description and builds the 4 KiB decoding table with the decode
 * kernel's own builders (cz_parse_sections, cz_huf_weight_table, cz_huf_rank_wave, cz_huf_fill), then the waves share the
 * streams.  A Treeless section rebuilds the tree of the block that defined it (the list entry says which): sections stay
 * independent of each other, so the 986 blocks of one large frame spread over the whole chip.
 *
 * A stream is decoded in TILES of 1 KiB of bitstream, 16 bytes per lane, staged in LDS by coalesced 16-byte loads one tile
 * ahead.  Prefix codes resynchronise, so within a tile
 *   1. every lane decodes, counting only, the symbols that START in its 16 bytes, from a guessed start (its upper boundary);
 *      the true start of lane i + 1 is where lane i ended: lanes whose start moved decode again until nothing moves
 *      (lane 0 starts where the tile above ended, so this is exact; two or three rounds in practice);
 *      the counting passes step two symbols per table lookup where the entry says so (cz_huf_fill_multi);
 *   2. a wave prefix sum of the counts gives every lane its place in the tile's output;
 *   3. the lanes decode once more and write their symbols, a byte each, into an LDS staging buffer, which the wave then
 *      writes out as whole aligned 16-byte pieces (a partial piece is carried to the next tile).
 * No global-memory access sits inside a decoding loop: bits come from LDS (one 64-bit window per five lookups), the table is
 * in LDS, symbols go to LDS.  The result is what the reference's sequential reader produces — or nothing: on ANY
 * irregularity (tree error, padding, a stream that does not end exactly, a symbol count that is not the ceil(regen / 4)
 * split, ...) the frame is marked "literals not done" (lit_first[f] = frame_pre[f] = 0) and cz_decode_frames_kernel does it
 * from scratch, reporting the reference's status.  Nothing here reports errors.
 *
 * cz_tile_kernel: Raw / RLE runs whose place is known from the headers (block_decoder.cairo:95-122): one workgroup of 256
 * threads per run (at most 128 KiB), taken from a shared counter, 16 bytes per thread and step, eight loads in flight per thread.
 */
#ifndef CZH_WAVES
#define CZH_WAVES 2                       /* waves per workgroup: wave w decodes streams w, w + CZH_WAVES, ... (two workgroups fit next to cz_chain_kernel's 135 KB of LDS) */
#endif
#define CZH_THREADS (64 * CZH_WAVES)
#define CZH_TILE 1024u                    /* bytes of bitstream per tile: 16 per lane */
#define CZH_STG 2048u                     /* symbols staged per flush (one tile of 5-bit codes yields about 1 640) */
#define CZH_WIN (CZH_STG - 16u)           /* symbols of a tile written per writing pass (the carried partial piece takes up to 15 more) */
#define CZH_LOOKUPS 5                     /* lookups per 64-bit window: 4 x 11 bits consumed + 11 bits of index */
struct CzHufWave {
    __attribute__((aligned(16))) uint8_t bits[16 + CZH_TILE + 16];       /* [0,16): the 16 bytes below the tile; the tile; 16 bytes of slack above */
    __attribute__((aligned(16))) uint8_t stg[CZH_STG + 64 + 32];         /* + the carried piece, + one dump byte per lane */
};
struct CzHufInfo { uint32_t seg, frame, regen, nstreams, fail, stream_off[4], stream_len[4], wave_fail[4]; };
__shared__ CzHufWave czh_w[CZH_WAVES];
__shared__ CzHufInfo czh_i;

/* 16 bytes at absolute address a: bytes outside the stream [S, E) read as zero, nothing outside [lo, hi) is touched */
__device__ static inline uint4 czh_load_chunk(uintptr_t a, uintptr_t S, uintptr_t E, uintptr_t lo, uintptr_t hi) {
    uint4 v; v.x = v.y = v.z = v.w = 0;
    if (a + 16 <= S || a >= E) return v;
    if (a >= lo && a + 16 <= hi) {
        __builtin_memcpy(&v, (CZ_GLOBAL const void*)a, 16);
        if (a < S) {                                                    /* zero the bytes below the stream start (the reader's zero extension, bit_reader_reverse.cairo:147-159) */
            const uint32_t nb = (uint32_t)(S - a);
            v.x = cz_mask_low_bytes(v.x, 0, nb); v.y = cz_mask_low_bytes(v.y, 4, nb); v.z = cz_mask_low_bytes(v.z, 8, nb); v.w = cz_mask_low_bytes(v.w, 12, nb);
        }
        return v;
    }
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t b = 0; b < 16; b++) { const uintptr_t q = a + b; if (q >= S && q < E) w[b >> 2] |= (uint32_t)(*(cz_gcptr)q) << (8 * (b & 3)); }
    v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
    return v;
}
/* 64 stream bits below buffer bit address g (exclusive), most significant first */
__device__ static inline uint64_t czh_window(const uint8_t* bits, uint32_t g) {
    const uint32_t wi = g >> 5, ph = g & 31u;
    const uint32_t* w = (const uint32_t*)bits;
    const uint32_t w2 = w[wi], w1 = w[wi - 1], w0 = w[wi - 2];
    return ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, ph) << 32) | __builtin_amdgcn_alignbit(w1, w0, ph);
}
/* The symbols that start above `stop`, from u (= unread bits of the stream, the next bit is bit u - 1) downwards.
 * COUNT: two symbols per lookup where the table says so; returns the count.  Otherwise: symbol k of the lane goes to
 * stg[at + k - lo] when lo <= at + k < hi (else to the lane's dump byte); `windowed` = 0 says every symbol is inside.
 * `boff`: buffer bit address of stream bit 0.
 * An interval is one 64-bit window and CZH_LOOKUPS lookups.  A lane that is more than 5 x max_bits above its `stop` runs the
 * interval without a test per lookup (five lookups consume at most that, so each of them — and the second symbol of a
 * double step — begins above `stop`); nearer to it, every lookup is tested. */
template <int COUNT>
__device__ static inline uint32_t czh_run(const uint8_t* bits, uint8_t* stg, int32_t boff, int32_t& u_, int32_t stop, int run, uint32_t mb, uint32_t at, uint32_t lo, uint32_t hi, int windowed) {
    int32_t u = u_; uint32_t n = 0;
    const uint32_t dump = CZH_STG + 16u + (uint32_t)LANE;
    const uint32_t shr = 32u - mb;
    for (;;) {
        const int live0 = run && u > stop;
        if (!__ballot(live0)) break;
        int32_t gb = boff + u;                                          /* (a lane that is done may stand anywhere: keep its reads inside the buffer) */
        gb = gb < 64 ? 64 : (gb > (int32_t)(8u * (16u + CZH_TILE)) ? (int32_t)(8u * (16u + CZH_TILE)) : gb);
        uint64_t buf = czh_window(bits, (uint32_t)gb);
        const int safe = live0 && u - stop > (int32_t)(mb * CZH_LOOKUPS);   /* a lookup consumes at most mb bits (both symbols of a double step lie inside the index bits) */
        if (__ballot(safe)) {
            if (safe) {
                if (COUNT) {
                    uint32_t used = 0;
#pragma unroll
                    for (int j = 0; j < CZH_LOOKUPS; j++) {
                        const uint32_t e = sh.a.huf[(uint32_t)(buf >> 32) >> shr];
                        const uint32_t l2 = e >> 12, nb = ((e >> 8) & 15u) + l2;
                        n += 1u + (l2 < 1u ? l2 : 1u);
                        buf <<= nb; used += nb;
                    }
                    u -= (int32_t)used;
                } else if (!windowed) {
                    uint8_t* const o = stg + at + n;
                    uint32_t used = 0;
#pragma unroll
                    for (int j = 0; j < CZH_LOOKUPS; j++) {
                        const uint32_t e = sh.a.huf[(uint32_t)(buf >> 32) >> shr];
                        const uint32_t nb = (e >> 8) & 15u;
                        o[j] = (uint8_t)e;
                        buf <<= nb; used += nb;
                    }
                    u -= (int32_t)used; n += CZH_LOOKUPS;
                } else {
#pragma unroll
                    for (int j = 0; j < CZH_LOOKUPS; j++) {
                        const uint32_t e = sh.a.huf[(uint32_t)(buf >> 32) >> shr];
                        const uint32_t nb = (e >> 8) & 15u, k = at + n;
                        stg[k >= lo && k < hi ? k - lo : dump] = (uint8_t)e;
                        n += 1u; buf <<= nb; u -= (int32_t)nb;
                    }
                }
            }
        }
        if (__ballot(live0 && !safe)) {
            const int careful = live0 && !safe;
            if (!COUNT && !windowed) {
#pragma unroll
                for (int j = 0; j < CZH_LOOKUPS; j++) {
                    const int live = careful && u > stop;
                    const uint32_t e = sh.a.huf[(uint32_t)(buf >> 32) >> shr];
                    const uint32_t nb = live ? (e >> 8) & 15u : 0u;
                    stg[live ? at + n : dump] = (uint8_t)e;
                    n += live ? 1u : 0u;
                    buf <<= nb; u -= (int32_t)nb;
                }
            } else
#pragma unroll
            for (int j = 0; j < CZH_LOOKUPS; j++) {
                const int live = careful && u > stop;
                const uint32_t e = sh.a.huf[(uint32_t)(buf >> 32) >> shr];
                const uint32_t l1 = (e >> 8) & 15u;
                uint32_t nb;
                if (COUNT) {
                    const uint32_t l2 = u - stop > (int32_t)mb ? e >> 12 : 0u;   /* the second symbol must begin above `stop` too */
                    nb = live ? l1 + l2 : 0u;
                    n += live ? (l2 ? 2u : 1u) : 0u;
                } else {
                    nb = live ? l1 : 0u;
                    const uint32_t k = at + n;
                    stg[live && k >= lo && k < hi ? k - lo : dump] = (uint8_t)e;
                    n += live ? 1u : 0u;
                }
                buf <<= nb; u -= (int32_t)nb;
            }
        }
    }
    u_ = u;
    return n;
}
/* staging -> global: bytes [from, upto) of stg correspond to g[from .. upto); whole 16-byte pieces go out as such (g and stg
   are 16-byte aligned), the ragged ends byte by byte */
__device__ static inline void czh_flush(const uint8_t* stg, cz_gptr g, uint32_t from, uint32_t upto) {
    const uint32_t c0 = from >> 4, c1 = (upto + 15u) >> 4;
    for (uint32_t c = c0 + (uint32_t)LANE; c < c1; c += 64u) {
        const uint32_t b0 = 16u * c;
        if (b0 >= from && b0 + 16u <= upto) *(cz_gptr4)(g + b0) = *(const uint4*)(stg + b0);
        else for (uint32_t b = b0 < from ? from : b0; b < b0 + 16u && b < upto; b++) g[b] = stg[b];
    }
}
/* One huff0 stream [S, S + len) -> exactly `cap` symbols at dst.  Returns 0, or 1 when the stream is not what that needs
   (the decode kernel then finds out what it is).  All 64 lanes of one wave. */
__device__ static int czh_decode_stream(CzHufWave& ws, uintptr_t S, uint32_t len, uintptr_t lo_safe, uintptr_t hi_safe, cz_gptr dst, uint32_t cap, uint32_t mb, int exact_end) {
    if (len == 0) return 1;
    const uintptr_t E = S + len;
    const uint32_t lastb = *(cz_gcptr)(E - 1);
    if (lastb == 0) return 1;                                           /* more than 8 padding bits (literals_section_decoder.cairo:190-207) */
    const int32_t P0 = (int32_t)len * 8 - (int32_t)(__clz((int)lastb) - 24 + 1);
    const uintptr_t top0 = (E + 15u) & ~(uintptr_t)15u;
    uint32_t carry = (uint32_t)((uintptr_t)dst & 15u), skip = carry;    /* bytes of the staging buffer's first piece that are not ours */
    cz_gptr g = dst - carry;
    uint32_t written = 0; int32_t u_carry = P0;
    uint8_t* const stg = ws.stg;
    /* tile 0 is loaded here, every later tile one tile ahead */
    uint4 nx = czh_load_chunk(top0 - 16u * ((uintptr_t)LANE + 1u), S, E, lo_safe, hi_safe);
    uint4 nbl = uint4{0, 0, 0, 0};
    if (LANE == 0) nbl = czh_load_chunk(top0 - CZH_TILE - 16u, S, E, lo_safe, hi_safe);
    for (uint32_t j = 0;; j++) {
        const uintptr_t Thi = top0 - (uintptr_t)CZH_TILE * j, LB = Thi - CZH_TILE - 16u;   /* LB: address of ws.bits[0] */
        cz_wave_sync();                                                 /* the tile above is no longer read */
        *(uint4*)&ws.bits[16u + CZH_TILE - 16u * ((uint32_t)LANE + 1u)] = nx;
        if (LANE == 0) { *(uint4*)&ws.bits[0] = nbl; *(uint4*)&ws.bits[16u + CZH_TILE] = uint4{0, 0, 0, 0}; }
        cz_wave_sync();
        {
            const uintptr_t Tn = Thi - CZH_TILE;                        /* top of the next tile */
            nx = czh_load_chunk(Tn - 16u * ((uintptr_t)LANE + 1u), S, E, lo_safe, hi_safe);
            if (LANE == 0) nbl = czh_load_chunk(Tn - CZH_TILE - 16u, S, E, lo_safe, hi_safe);
        }
        const int32_t boff = (int32_t)((intptr_t)S - (intptr_t)LB) * 8;  /* buffer bit address of stream bit 0 (negative when the stream begins below the buffer) */
        const int32_t qhi = (int32_t)((intptr_t)Thi - 16 * (intptr_t)LANE - (intptr_t)S) * 8, qlo = qhi - 128;
        const int32_t stop = qlo > 0 ? qlo : 0;
        /* 1. counting rounds */
        int32_t s = LANE == 0 ? u_carry : (qhi < P0 ? qhi : P0), e = s; uint32_t n = 0;
        int changed = 1;
        for (int round = 0; round < 66; round++) {
            if (__ballot(changed)) {
                int32_t u = s;
                const uint32_t got = czh_run<1>(ws.bits, stg, boff, u, stop, changed, mb, 0, 0, 0, 0);
                if (changed) { n = got; e = u; }
            }
            int32_t pe = __shfl_up(e, 1u);
            if (LANE == 0) pe = u_carry;
            changed = pe != s;
            if (changed) s = pe;
            if (!__ballot(changed)) break;
        }
        /* 2. places */
        const uint32_t incl = cz_wave_incl_scan(n), T = cz_readlane(incl, 63), at = incl - n;
        if (written + T > cap) return 1;
        /* 3. writing passes (one, unless the tile holds more symbols than the staging buffer) and flushes */
        for (uint32_t wlo = 0; wlo < T; wlo += CZH_WIN) {
            const uint32_t cnt = T - wlo < CZH_WIN ? T - wlo : CZH_WIN;
            int32_t u = s;
            cz_wave_sync();
            czh_run<0>(ws.bits, stg + carry, boff, u, stop, 1, mb, at, wlo, wlo + cnt, T > CZH_WIN);
            cz_wave_sync();
            const uint32_t have = carry + cnt, full = have & ~15u;
            if (full) {
                czh_flush(stg, g, skip, full);
                cz_wave_sync();
                uint8_t t = 0;
                if ((uint32_t)LANE < have - full) t = stg[full + (uint32_t)LANE];
                cz_wave_sync();
                if ((uint32_t)LANE < have - full) stg[LANE] = t;
                g += full; skip = 0;
            }
            carry = have - full;
        }
        written += T;
        u_carry = cz_unii(__shfl(e, 63));
        if (u_carry <= 0) break;
    }
    cz_wave_sync();
    if (carry > skip) czh_flush(stg, g, skip, carry);
    if (written != cap) return 1;                                       /* literals_section_decoder.cairo:172-178, and the ceil(regen / 4) split this kernel relies on */
    if (exact_end && u_carry != 0) return 1;                            /* :234-241 */
    return 0;
}

/* wave 0: header and tree of the block at `blk` (staged head first); leaves the table in sh.a.huf, the sizes in sh.bc.
   Returns the status of cz_parse_sections (uniform). */
__device__ static int czh_parse_block(cz_gcptr blk, uint32_t bsize) {
    CzBroadcast& bc = sh.bc;
    const uint32_t stage_hi = bsize < 512 ? bsize : 512;
    cz_wave_sync();
    for (uint32_t i = (uint32_t)LANE; i < stage_hi; i += 64) sh.a.t1.stage[i] = blk[i];
    cz_wave_sync();
    if (LANE == 0) bc.err = cz_parse_sections(blk, bsize, stage_hi, 0, 0);
    cz_wave_sync();
    if (cz_unii(bc.err) == CZ_PARSE_NEED_WTAB) {
        cz_huf_weight_table();
        cz_wave_sync();
        if (LANE == 0) bc.err = cz_parse_sections(blk, bsize, stage_hi, 0, 1);
        cz_wave_sync();
    }
    const int e = cz_unii(bc.err);
    cz_wave_sync();
    if (e) return e;
    if (cz_uni(bc.huf_fill)) {
        cz_huf_rank_wave(cz_uni(bc.huf_nsym)); cz_wave_sync();
        cz_huf_fill(cz_uni(bc.huf_nsym)); cz_wave_sync();
        cz_huf_fill_multi(); cz_wave_sync();
    }
    return 0;
}

/* A section the huff0 kernels will not do: the whole frame goes back to cz_decode_frames_kernel, which does it from scratch (no
   literals, no leading blocks: lit_first[f] = 0 and the count and the "regular" / "done" marks of frame_pre[f] cleared; the marks of
   the execute stage — listed for cz_wexec_kernel, claimed, on the fall-back list — stay, so that exactly one kernel owns the frame
   from here on) and listed here, once however many of its sections fail.  One lane calls it. */
__device__ static inline void czh_hand_back(const cz_batch_args& a, uint32_t f) {
    a.lit_first[f] = 0;
    atomicAnd(&a.frame_pre[f], ~(CZ_PRE_REGULAR | CZ_PRE_DONE | CZ_PRE_COUNT));
    cz_list_fallback(a, f);
}

extern "C" __global__ void __launch_bounds__(CZH_THREADS, CZH_WAVES == 2 ? 6 : 8) cz_huf_kernel(cz_batch_args a) {
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t nseg = 0; for (int c = 0; c < 20; c++) nseg += a.scan_ctl[136 + c];
    if (nseg > a.lit_seg_capacity) nseg = a.lit_seg_capacity;
    CzBroadcast& bc = sh.bc;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) { czh_i.seg = atomicAdd(&a.scan_ctl[200], 1u); czh_i.fail = 0; for (int k = 0; k < 4; k++) czh_i.wave_fail[k] = 0; }
        __syncthreads();
        const uint32_t si = cz_uni(czh_i.seg);
        if (si >= nseg) break;
        const cz_lit_seg sg = a.lit_segs[si];
        if (sg.frame == 0xFFFFFFFFu) continue;                          /* void entry: its frame is not pre-passed */
        const uint32_t f = cz_uni(sg.frame);
        cz_gcptr frame = (cz_gcptr)(a.in_base + a.in_off[f]); const uint64_t flen = a.in_len[f];
        cz_gcptr blk = frame + cz_uni(sg.blk_off); const uint32_t bsize = cz_uni(sg.bsize);
        if (wave == 0) {
            int bad = 0;
            if (LANE == 0) sh.huf_max_bits = 0;
            cz_wave_sync();
            if (cz_uni(sg.def) != 0xFFFFFFFFu) {
                /* Treeless: sizes of this block first (they end up in czh_i), then the tree of the block that defined it */
                const cz_lit_seg dg = a.lit_segs[cz_uni(sg.def) < nseg ? cz_uni(sg.def) : 0];
                if (cz_uni(sg.def) >= nseg || cz_uni(dg.frame) != f) bad = 1;
                else {
                    if (LANE == 0) sh.huf_max_bits = 1;                 /* "a table exists" for the parser; the real one follows */
                    bad = czh_parse_block(blk, bsize) != 0 || cz_uni(bc.lit_type) != 3;
                    if (LANE == 0) { czh_i.regen = bc.regen; czh_i.nstreams = bc.nstreams; for (int k = 0; k < 4; k++) { czh_i.stream_off[k] = bc.stream_off[k]; czh_i.stream_len[k] = bc.stream_len[k]; } }
                    cz_wave_sync();
                    if (LANE == 0) sh.huf_max_bits = 0;
                    cz_wave_sync();
                    if (!bad) bad = czh_parse_block(frame + cz_uni(dg.blk_off), cz_uni(dg.bsize)) != 0 || cz_uni(bc.lit_type) != 2;
                }
            } else {
                bad = czh_parse_block(blk, bsize) != 0 || cz_uni(bc.lit_type) != 2;
                if (LANE == 0) { czh_i.regen = bc.regen; czh_i.nstreams = bc.nstreams; for (int k = 0; k < 4; k++) { czh_i.stream_off[k] = bc.stream_off[k]; czh_i.stream_len[k] = bc.stream_len[k]; } }
            }
            cz_wave_sync();
            if (LANE == 0 && (bad || czh_i.regen != sg.regen || sh.huf_max_bits == 0)) czh_i.fail = 1;
        }
        __syncthreads();
        if (!cz_uni(czh_i.fail)) {
            const uint32_t regen = cz_uni(czh_i.regen), ns = cz_uni(czh_i.nstreams);
            const uint32_t seg = (regen + 3) >> 2;
            if (ns == 4 && 3 * seg > regen) { if (threadIdx.x == 0) czh_i.fail = 1; }   /* (a split that cannot be: the decode kernel sorts it out) */
            else {
#pragma unroll 1
                for (uint32_t k = wave; k < ns; k += CZH_WAVES) {
                    /* (everything the call needs is read from LDS here: nothing but k lives across it) */
                    const uint32_t regen_ = cz_uni(czh_i.regen), seg_ = (regen_ + 3) >> 2, ns_ = cz_uni(czh_i.nstreams);
                    const uint32_t cap = ns_ == 4 ? (k < 3 ? seg_ : regen_ - 3 * seg_) : regen_;
                    cz_gptr tg = (cz_gptr)((sg.direct ? a.out_base : a.lit_arena) + sg.dst) + (ns_ == 4 ? k * seg_ : 0u);
                    const int r = czh_decode_stream(czh_w[wave], (uintptr_t)(blk + cz_uni(czh_i.stream_off[k])), cz_uni(czh_i.stream_len[k]),
                                                    (uintptr_t)frame, (uintptr_t)(frame + flen), tg, cap, cz_uni(sh.huf_max_bits), ns_ == 4);
                    if (r && LANE == 0) czh_i.wave_fail[k] = 1;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && (czh_i.fail | czh_i.wave_fail[0] | czh_i.wave_fail[1] | czh_i.wave_fail[2] | czh_i.wave_fail[3])) {
            czh_hand_back(a, f);
        }
    }
}

/* cz_huf1_kernel: the same list, ONE wave per section, with the decode kernel's own stream decoder (cz_decode_huf_literals:
 * sixteen bit ranges per stream, one lane each, bits straight from global memory).  Per symbol it does less work than
 * cz_huf_kernel and it needs 5.5 KB of LDS, so four of its workgroups fit on a CU next to cz_chain_kernel's 135 KB — it is the
 * form that runs WHILE the chain kernel runs (cz_huf_kernel's workgroups would get one CU slot in eight there).  It stops taking
 * sections once the chain kernel's waves have all counted themselves out; cz_huf_kernel, launched behind the chain kernel,
 * finishes the list with the whole chip.  A batch without chains leaves everything to cz_huf_kernel. */
extern "C" __global__ void __launch_bounds__(CZ_WG_THREADS, 4) cz_huf1_kernel(cz_batch_args a) {
    uint32_t nseg = 0, nchain = 0;
    for (int c = 0; c < 20; c++) { nseg += a.scan_ctl[136 + c]; nchain += a.scan_ctl[c]; }
    if (nseg > a.lit_seg_capacity) nseg = a.lit_seg_capacity;
    if (!nchain) return;
    CzBroadcast& bc = sh.bc;
    for (;;) {
        __syncthreads();
        if (LANE == 0) {
            const uint32_t out = *(volatile uint32_t*)&a.scan_ctl[205];
            sh.frame_idx = out >= a.chain_grid ? 0xFFFFFFFFu : atomicAdd(&a.scan_ctl[200], 1u);
        }
        __syncthreads();
        const uint32_t si = cz_uni(sh.frame_idx);
        if (si >= nseg) break;
        const cz_lit_seg sg = a.lit_segs[si];
        if (sg.frame == 0xFFFFFFFFu) continue;
        const uint32_t f = cz_uni(sg.frame);
        cz_gcptr frame = (cz_gcptr)(a.in_base + a.in_off[f]);
        cz_gcptr blk = frame + cz_uni(sg.blk_off); const uint32_t bsize = cz_uni(sg.bsize);
        int bad = 0;
        if (LANE == 0) sh.huf_max_bits = 0;
        cz_wave_sync();
        if (cz_uni(sg.def) != 0xFFFFFFFFu) {
            /* Treeless: sizes of this block first, then the tree of the block that defined it (the table shares its LDS with the
               parser's stage), then the sizes back into the broadcast slots the stream decoder reads */
            const cz_lit_seg dg = a.lit_segs[cz_uni(sg.def) < nseg ? cz_uni(sg.def) : 0];
            if (cz_uni(sg.def) >= nseg || cz_uni(dg.frame) != f) bad = 1;
            else {
                if (LANE == 0) sh.huf_max_bits = 1;
                bad = czh_parse_block(blk, bsize) != 0 || cz_uni(bc.lit_type) != 3;
                const uint32_t regen = cz_uni(bc.regen), ns = cz_uni(bc.nstreams);
                const uint32_t o0 = cz_uni(bc.stream_off[0]), o1 = cz_uni(bc.stream_off[1]), o2 = cz_uni(bc.stream_off[2]), o3 = cz_uni(bc.stream_off[3]);
                const uint32_t l0 = cz_uni(bc.stream_len[0]), l1 = cz_uni(bc.stream_len[1]), l2 = cz_uni(bc.stream_len[2]), l3 = cz_uni(bc.stream_len[3]);
                cz_wave_sync();
                if (LANE == 0) sh.huf_max_bits = 0;
                cz_wave_sync();
                if (!bad) bad = czh_parse_block(frame + cz_uni(dg.blk_off), cz_uni(dg.bsize)) != 0 || cz_uni(bc.lit_type) != 2;
                cz_wave_sync();
                if (LANE == 0) { bc.regen = regen; bc.nstreams = ns; bc.stream_off[0] = o0; bc.stream_off[1] = o1; bc.stream_off[2] = o2; bc.stream_off[3] = o3;
                                 bc.stream_len[0] = l0; bc.stream_len[1] = l1; bc.stream_len[2] = l2; bc.stream_len[3] = l3; }
                cz_wave_sync();
            }
        } else bad = czh_parse_block(blk, bsize) != 0 || cz_uni(bc.lit_type) != 2;
        if (!bad && (cz_uni(bc.regen) != cz_uni(sg.regen) || cz_uni(sh.huf_max_bits) == 0)) bad = 1;
        __syncthreads();
        if (!bad) bad = cz_decode_huf_literals(blk, (cz_gptr)((sg.direct ? a.out_base : a.lit_arena) + sg.dst)) != 0;
        __syncthreads();
        if (bad && LANE == 0) {
            czh_hand_back(a, f);
        }
    }
}

/* Raw / RLE runs.  One workgroup per run; the list is in no particular order (runs are at most 128 KiB). */
/* written once, read by another kernel later: keep the lines out of the caches' way */
#ifdef CZ_EMU
#define CZT_STORE(p, v) (*(cz_gptr4)(p) = (v))
#else
typedef uint32_t czt_u4 __attribute__((ext_vector_type(4)));
#define CZT_STORE(p, v) do { const uint4 t_ = (v); czt_u4 n_; n_.x = t_.x; n_.y = t_.y; n_.z = t_.z; n_.w = t_.w; __builtin_nontemporal_store(n_, (CZ_GLOBAL czt_u4*)(p)); } while (0)
#endif
extern "C" __global__ void __launch_bounds__(256) cz_tile_kernel(cz_batch_args a) {
    uint32_t nseg = a.scan_ctl[201];
    if (nseg > a.copy_seg_capacity) nseg = a.copy_seg_capacity;
    __shared__ uint32_t czt_next;
    for (;;) {
        /* runs come off a shared counter: their cost differs (a fill only writes), a fixed stride would leave half the workgroups
           with the cheap half */
        __syncthreads();
        if (threadIdx.x == 0) czt_next = atomicAdd(&a.scan_ctl[203], 1u);
        __syncthreads();
        const uint32_t si = cz_uni(czt_next);
        if (si >= nseg) break;
        const cz_copy_seg sg = a.copy_segs[si];
        const uint32_t n = sg.len;
        if (!n) continue;
        cz_gptr dst = (cz_gptr)(a.out_base + sg.dst); cz_gcptr src = (cz_gcptr)(a.in_base + sg.src);
        const uint32_t t = threadIdx.x;
        /* head up to the first 16-byte boundary of the destination, then 16 bytes per thread and step, then the tail */
        uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
        if (head > n) head = n;
        const uint32_t nvec = (n - head) >> 4, tail0 = head + (nvec << 4);
        if (sg.fill) {
            const uint32_t b = src[0], w = 0x01010101u * b;
            const uint4 v = uint4{w, w, w, w};
            if (t < head) dst[t] = (uint8_t)b;
            for (uint32_t i = t; i < nvec; i += 256u) CZT_STORE(dst + head + 16u * i, v);
            if (t < n - tail0) dst[tail0 + t] = (uint8_t)b;
        } else {
            if (t < head) dst[t] = src[t];
            cz_gcptr s = src + head; cz_gptr d = dst + head;
            uint32_t i = t;
            for (; i + 7u * 256u < nvec; i += 8u * 256u) {              /* eight 16-byte loads in flight per thread (the source need not be aligned) */
                uint4 v0, v1, v2, v3, v4, v5, v6, v7;
                __builtin_memcpy(&v0, s + 16u * i, 16); __builtin_memcpy(&v1, s + 16u * (i + 256u), 16);
                __builtin_memcpy(&v2, s + 16u * (i + 512u), 16); __builtin_memcpy(&v3, s + 16u * (i + 768u), 16);
                __builtin_memcpy(&v4, s + 16u * (i + 1024u), 16); __builtin_memcpy(&v5, s + 16u * (i + 1280u), 16);
                __builtin_memcpy(&v6, s + 16u * (i + 1536u), 16); __builtin_memcpy(&v7, s + 16u * (i + 1792u), 16);
                CZT_STORE(d + 16u * i, v0); CZT_STORE(d + 16u * (i + 256u), v1); CZT_STORE(d + 16u * (i + 512u), v2); CZT_STORE(d + 16u * (i + 768u), v3);
                CZT_STORE(d + 16u * (i + 1024u), v4); CZT_STORE(d + 16u * (i + 1280u), v5); CZT_STORE(d + 16u * (i + 1536u), v6); CZT_STORE(d + 16u * (i + 1792u), v7);
            }
            for (; i < nvec; i += 256u) { uint4 v; __builtin_memcpy(&v, s + 16u * i, 16); CZT_STORE(d + 16u * i, v); }
            if (t < n - tail0) dst[tail0 + t] = src[tail0 + t];
        }
    }
}
