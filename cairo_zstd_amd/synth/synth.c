/*
 * synth.c — deterministic synthetic zstd frame generator for the benchmark / parity
 * workloads of BASELINE.md (configs 2, 3, 4a, 4b, 5).
 *
 * The GPU box receives only this repository, and 10 K – 100 K x 128 KiB inputs cannot be
 * committed, so the inputs are generated there.  This is a purpose-built *encoder* for
 * controlled block shapes (it does no match finding): Huffman literals (direct or
 * FSE-compressed tree description, 1 or 4 streams, Treeless re-use), sequences with
 * Predefined / RLE / FSE_Compressed / Repeat tables, Raw and RLE blocks, multi-block frames.
 * Its output is validated against the system libzstd in the build container
 * (tests/test_synth.py) and decodes identically under the CPU oracle.
 *
 * It is neither part of the decode product path nor of the oracle.
 * Seeds: splitmix64, one stream per frame = seed ^ (frame_index * golden) so the batch is
 * independent of the number of generator threads.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define CZS_API __attribute__((visibility("default")))

enum { CZS_RAW_RLE = 2, CZS_HUF_LITERALS = 3, CZS_FULL_4A = 4, CZS_FULL_4B = 41, CZS_MIX = 5 };

/* ---------------------------------------------------------------- rng */
typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t* r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
static inline uint32_t rng_below(rng_t* r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }

/* --------------------------------------------------------- bit writer */
typedef struct { uint8_t* p; size_t cap; size_t nbits; int overflow; } bw_t;
static void bw_init(bw_t* w, uint8_t* p, size_t cap) { w->p = p; w->cap = cap; w->nbits = 0; w->overflow = 0; memset(p, 0, cap); }
static inline void bw_add(bw_t* w, uint64_t v, unsigned n) {
    if (!n) return;
    if ((w->nbits + n + 7) / 8 + 8 > w->cap) { w->overflow = 1; return; }
    v &= (n >= 64) ? ~0ULL : ((1ULL << n) - 1);
    size_t byte = w->nbits >> 3; unsigned sh = w->nbits & 7;
    uint64_t cur; memcpy(&cur, w->p + byte, 8);
    cur |= v << sh; memcpy(w->p + byte, &cur, 8);
    if (sh + n > 64) w->p[byte + 8] |= (uint8_t)(v >> (64 - sh));
    w->nbits += n;
}
/* close a reversed stream: final 1 marker, zero padding to the byte */
static size_t bw_close_reversed(bw_t* w) { bw_add(w, 1, 1); return (w->nbits + 7) / 8; }
static size_t bw_bytes(const bw_t* w) { return (w->nbits + 7) / 8; }

static inline unsigned highbit32(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }

/* ----------------------------------------------------------------- FSE */
typedef struct {
    int log; int nsym;                 /* alphabet size = max symbol + 1 */
    int16_t norm[256];
    uint16_t base[512]; uint8_t nb[512], sym[512];  /* decoder view */
    uint16_t first_state[256];         /* a state of the symbol with maximal nbits */
    uint16_t* map;                     /* map[sym*size + S] = state s with symbol sym whose interval holds S */
} fse_t;

static void fse_build(fse_t* t) {
    const uint32_t size = 1u << t->log;
    uint32_t hi = size;
    for (int s = 0; s < t->nsym; s++) if (t->norm[s] == -1) { hi--; t->sym[hi] = (uint8_t)s; t->base[hi] = 0; t->nb[hi] = (uint8_t)t->log; }
    uint32_t pos = 0, step = (size >> 1) + (size >> 3) + 3;
    for (int s = 0; s < t->nsym; s++)
        for (int j = 0; j < t->norm[s]; j++) {
            t->sym[pos] = (uint8_t)s;
            do { pos = (pos + step) & (size - 1); } while (pos >= hi);
        }
    uint32_t cnt[256]; memset(cnt, 0, sizeof cnt);
    for (uint32_t i = 0; i < hi; i++) {
        uint32_t s = t->sym[i], n = (uint32_t)t->norm[s], k = cnt[s]++;
        uint32_t mask = 1u << highbit32(n), slices = (mask == n) ? n : mask * 2;
        uint32_t dbl = slices - n, single = n - dbl, width = size / slices, nbits = highbit32(width);
        if (k < dbl) { t->base[i] = (uint16_t)(single * width + k * width * 2); t->nb[i] = (uint8_t)(nbits + 1); }
        else { t->base[i] = (uint16_t)((k - dbl) * width); t->nb[i] = (uint8_t)nbits; }
    }
    t->map = (uint16_t*)malloc((size_t)t->nsym * size * sizeof(uint16_t));
    uint8_t best[256]; memset(best, 0, sizeof best);
    for (int s = 0; s < t->nsym; s++) t->first_state[s] = 0xFFFF;
    for (uint32_t i = 0; i < size; i++) {
        uint32_t s = t->sym[i];
        for (uint32_t S = t->base[i]; S < (uint32_t)t->base[i] + (1u << t->nb[i]); S++) t->map[s * size + S] = (uint16_t)i;
        if (t->first_state[s] == 0xFFFF || t->nb[i] > best[s]) { t->first_state[s] = (uint16_t)i; best[s] = t->nb[i]; }
    }
}
static void fse_release(fse_t* t) { free(t->map); t->map = NULL; }

/* counts -> normalized counts summing to 2^log (every present symbol >= 1) */
static void fse_normalize(fse_t* t, const uint32_t* count, int nsym, int log) {
    uint64_t total = 0; for (int s = 0; s < nsym; s++) total += count[s];
    const int size = 1 << log; int sum = 0;
    t->log = log; t->nsym = nsym;
    for (int s = 0; s < nsym; s++) {
        if (!count[s]) { t->norm[s] = 0; continue; }
        int64_t p = (int64_t)(((uint64_t)count[s] * (uint64_t)size + total / 2) / total);
        if (p < 1) p = 1;
        t->norm[s] = (int16_t)p; sum += (int)p;
    }
    /* fix the sum on the largest symbols */
    while (sum != size) {
        int best = -1;
        for (int s = 0; s < nsym; s++) if (t->norm[s] > (sum > size ? 1 : 0) && (best < 0 || t->norm[s] > t->norm[best])) best = s;
        int d = size - sum;
        if (d < 0 && -d >= t->norm[best]) d = -(t->norm[best] - 1);
        t->norm[best] = (int16_t)(t->norm[best] + d); sum += d;
    }
}
/* table description, format of fse_decoder.cairo:258-368 / zstd FSE_writeNCount */
static void fse_write_ncount(bw_t* w, const fse_t* t) {
    bw_add(w, (uint64_t)(t->log - 5), 4);
    int remaining = 1 << t->log; int s = 0;
    while (remaining > 0) {
        int prob = t->norm[s++];
        uint32_t maxv = (uint32_t)remaining + 1, v = (uint32_t)(prob + 1);
        unsigned bits = highbit32(maxv) + 1;
        uint32_t low = (1u << bits) - 1 - maxv, mask = (1u << (bits - 1)) - 1;
        if (v < low) bw_add(w, v, bits - 1);
        else if (v <= mask) bw_add(w, v, bits);
        else bw_add(w, v + low, bits);
        remaining -= prob < 0 ? 1 : prob;
        if (prob == 0) {
            int extra = 0;
            while (s < t->nsym && t->norm[s] == 0) { extra++; s++; }
            while (extra >= 3) { bw_add(w, 3, 2); extra -= 3; }
            bw_add(w, (uint64_t)extra, 2);
        }
    }
    w->nbits = (w->nbits + 7) & ~(size_t)7;
}

/* ------------------------------------------------------------- Huffman */
typedef struct { uint8_t len[256]; uint16_t code[256]; uint8_t weight[256]; int maxbits; int last; /* highest present symbol */ } huf_t;

/* length-limited (<= 11) Huffman lengths from counts */
static int huf_build(huf_t* h, const uint32_t* count) {
    int n = 0, idx[256]; memset(h, 0, sizeof *h);
    for (int s = 0; s < 256; s++) if (count[s]) idx[n++] = s;
    if (n < 2) return 1;
    /* simple O(n^2) Huffman */
    uint64_t w[512]; int parent[512], alive[512], m = n;
    for (int i = 0; i < n; i++) { w[i] = count[idx[i]]; parent[i] = -1; alive[i] = 1; }
    for (int r = 0; r < n - 1; r++) {
        int a = -1, b = -1;
        for (int i = 0; i < m; i++) if (alive[i]) { if (a < 0 || w[i] < w[a]) { b = a; a = i; } else if (b < 0 || w[i] < w[b]) b = i; }
        w[m] = w[a] + w[b]; parent[m] = -1; alive[m] = 1; parent[a] = parent[b] = m; alive[a] = alive[b] = 0; m++;
    }
    int len[256];
    for (int i = 0; i < n; i++) { int d = 0; for (int p = parent[i]; p >= 0; p = parent[p]) d++; len[i] = d > 11 ? 11 : d; }
    /* Kraft repair at limit 11 */
    int64_t K = 0; for (int i = 0; i < n; i++) K += 1 << (11 - len[i]);
    while (K > 2048) {              /* lengthen the longest code that is still < 11 */
        int best = -1; for (int i = 0; i < n; i++) if (len[i] < 11 && (best < 0 || len[i] > len[best])) best = i;
        K -= 1 << (11 - len[best] - 1); len[best]++;
    }
    while (K < 2048) {              /* shorten where it fits */
        int best = -1; for (int i = 0; i < n; i++) if (len[i] > 1 && (1 << (11 - len[i])) <= 2048 - K && (best < 0 || len[i] < len[best])) best = i;
        if (best < 0) return 2;
        K += 1 << (11 - len[best]); len[best]--;
    }
    int maxlen = 0; for (int i = 0; i < n; i++) if (len[i] > maxlen) maxlen = len[i];
    h->maxbits = maxlen; h->last = idx[n - 1];
    for (int i = 0; i < n; i++) { h->len[idx[i]] = (uint8_t)len[i]; h->weight[idx[i]] = (uint8_t)(maxlen + 1 - len[i]); }
    /* canonical codes as the decoder lays them out (huff0_decoder.cairo:410-467): longest
       codes first, ascending symbol within a length */
    uint32_t rank_count[13] = {0}, next[13];
    for (int s = 0; s < 256; s++) if (h->len[s]) rank_count[h->len[s]]++;
    uint32_t pos = 0;
    for (int b = maxlen; b >= 1; b--) { next[b] = pos; pos += rank_count[b] << (maxlen - b); }
    for (int s = 0; s < 256; s++) if (h->len[s]) { int b = h->len[s]; h->code[s] = (uint16_t)(next[b] >> (maxlen - b)); next[b] += 1u << (maxlen - b); }
    return 0;
}
/* tree description.  mode 0: direct 4-bit weights, 1: FSE-compressed, 2: whichever fits
   (FSE preferred).  Returns bytes written or 0 when impossible. */
static size_t huf_write_tree(const huf_t* h, uint8_t* out, size_t cap, int mode) {
    int nw = h->last;              /* weights for symbols 0..last-1, last one implied */
    if (nw < 1) return 0;
    if (mode == 1 || mode == 2) {
        uint32_t cnt[16] = {0}; for (int i = 0; i < nw; i++) cnt[h->weight[i]]++;
        int distinct = 0, maxw = 0; for (int i = 0; i < 13; i++) if (cnt[i]) { distinct++; maxw = i; }
        if (distinct >= 2 && nw >= 2) {
            fse_t t; fse_normalize(&t, cnt, maxw + 1, 6); fse_build(&t);
            uint8_t tmp[512]; bw_t w; bw_init(&w, tmp, sizeof tmp);
            fse_write_ncount(&w, &t);
            size_t hdr = bw_bytes(&w);
            uint8_t tmp2[512]; bw_t s; bw_init(&s, tmp2, sizeof tmp2);
            /* two interleaved states (huff0_decoder.cairo:227-274); see DESIGN.md "synth" */
            uint16_t S[256]; const uint32_t size = 1u << t.log;
            S[nw - 1] = t.first_state[h->weight[nw - 1]];
            S[nw - 2] = t.first_state[h->weight[nw - 2]];
            int ok = t.nb[S[nw - 2]] > 0;
            for (int i = nw - 3; i >= 0; i--) S[i] = t.map[h->weight[i] * size + S[i + 2]];
            for (int i = nw - 3; i >= 0; i--) bw_add(&s, (uint64_t)(S[i + 2] - t.base[S[i]]), t.nb[S[i]]);
            bw_add(&s, S[1], (unsigned)t.log); bw_add(&s, S[0], (unsigned)t.log);
            size_t body = bw_close_reversed(&s);
            fse_release(&t);
            if (ok && hdr + body <= 127 && 1 + hdr + body <= cap) {
                out[0] = (uint8_t)(hdr + body); memcpy(out + 1, tmp, hdr); memcpy(out + 1 + hdr, tmp2, body);
                return 1 + hdr + body;
            }
        }
        if (mode == 1) return 0;
    }
    if (nw > 128) return 0;
    size_t need = 1 + (size_t)(nw + 1) / 2; if (need > cap) return 0;
    out[0] = (uint8_t)(127 + nw);
    memset(out + 1, 0, need - 1);
    for (int i = 0; i < nw; i++) out[1 + i / 2] |= (i & 1) ? h->weight[i] : (uint8_t)(h->weight[i] << 4);
    return need;
}
/* one reversed Huffman stream */
static size_t huf_encode_stream(const huf_t* h, const uint8_t* lit, size_t n, uint8_t* out, size_t cap) {
    bw_t w; bw_init(&w, out, cap);
    for (size_t i = n; i-- > 0;) bw_add(&w, h->code[lit[i]], h->len[lit[i]]);
    size_t r = bw_close_reversed(&w);
    return w.overflow ? 0 : r;
}

/* ---------------------------------------------------- literals section */
/* type: 0 raw, 1 rle, 2 huffman, 3 treeless.  Returns bytes written (0 = failed). */
static size_t write_literals_section(uint8_t* out, size_t cap, int type, const uint8_t* lit, uint32_t n, const huf_t* h,
                                     int streams, int tree_mode) {
    if (type == 0 || type == 1) {
        size_t hl;
        if (n < 32) { out[0] = (uint8_t)(type | (n << 3)); hl = 1; }
        else if (n < 4096) { out[0] = (uint8_t)(type | (1 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); hl = 2; }
        else { out[0] = (uint8_t)(type | (3 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); out[2] = (uint8_t)(n >> 12); hl = 3; }
        if (type == 1) { out[hl] = lit[0]; return hl + 1; }
        if (hl + n > cap) return 0;
        memcpy(out + hl, lit, n); return hl + n;
    }
    uint8_t* body = (uint8_t*)malloc(2 * (size_t)n + 1024 + 64); size_t bl = 0, bcap = 2 * (size_t)n + 1024;
    if (type == 2) { bl = huf_write_tree(h, body, 200, tree_mode); if (!bl) { free(body); return 0; } }
    if (streams == 4) {
        size_t seg = (n + 3) / 4, off[5] = {0, seg, 2 * seg, 3 * seg, n};
        if (3 * seg > n) { free(body); return 0; }
        uint8_t* jt = body + bl; bl += 6;
        for (int k = 0; k < 4; k++) {
            size_t r = huf_encode_stream(h, lit + off[k], off[k + 1] - off[k], body + bl, bcap - bl);
            if (!r || (k < 3 && r > 65535)) { free(body); return 0; }
            if (k < 3) { jt[2 * k] = (uint8_t)r; jt[2 * k + 1] = (uint8_t)(r >> 8); }
            bl += r;
        }
    } else {
        size_t r = huf_encode_stream(h, lit, n, body + bl, bcap - bl);
        if (!r) { free(body); return 0; }
        bl += r;
    }
    size_t hl; uint32_t c = (uint32_t)bl;
    if (streams == 1) { if (n >= 1024 || c >= 1024) { free(body); return 0; }
        out[0] = (uint8_t)(type | (0 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)((n >> 4) | ((c & 3) << 6)); out[2] = (uint8_t)(c >> 2); hl = 3; }
    else if (n < 1024 && c < 1024) { out[0] = (uint8_t)(type | (1 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)((n >> 4) | ((c & 3) << 6)); out[2] = (uint8_t)(c >> 2); hl = 3; }
    else if (n < 16384 && c < 16384) { out[0] = (uint8_t)(type | (2 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); out[2] = (uint8_t)((n >> 12) | ((c & 63) << 2)); out[3] = (uint8_t)(c >> 6); hl = 4; }
    else { if (n >= (1u << 18) || c >= (1u << 18)) { free(body); return 0; }
        out[0] = (uint8_t)(type | (3 << 2) | ((n & 15) << 4)); out[1] = (uint8_t)(n >> 4); out[2] = (uint8_t)((n >> 12) | ((c & 3) << 6)); out[3] = (uint8_t)(c >> 2); out[4] = (uint8_t)(c >> 10); hl = 5; }
    if (hl + bl > cap) { free(body); return 0; }
    memcpy(out + hl, body, bl); free(body);
    return hl + bl;
}

/* ---------------------------------------------------- sequences section */
static const uint32_t LL_BASE[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
static const uint8_t LL_BITS[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const uint32_t ML_BASE[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
static const uint8_t ML_BITS[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
static const int16_t LL_DEFAULT[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const int16_t OF_DEFAULT[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const int16_t ML_DEFAULT[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
static inline int ll_code(uint32_t v) { int c = 35; while (LL_BASE[c] > v) c--; return c; }
static inline int ml_code(uint32_t v) { int c = 52; while (ML_BASE[c] > v) c--; return c; }

typedef struct { uint32_t ll, ml, ofv; } seq_t;            /* ofv = offset_value (1..3 repcode, else actual+3) */
typedef struct { fse_t t; int valid; int rle; } seq_table; /* encoder-side carried table (Repeat mode) */

static void set_default(fse_t* t, const int16_t* d, int n, int log) { t->log = log; t->nsym = n; memcpy(t->norm, d, (size_t)n * 2); fse_build(t); }

/* Picks / builds the table for one of LL, OF, ML.  mode: 0 predefined, 1 rle, 2 fse, 3 repeat.
   Writes the table description (if any) to w.  codes[] are the symbols to be coded. */
static int prepare_table(seq_table* st, int mode, const uint8_t* codes, uint32_t n, int nsym_max, int maxlog,
                         const int16_t* def, int defn, int deflog, bw_t* w, int sprinkle_low, rng_t* r) {
    if (mode == 3) return st->valid ? 0 : 1;
    if (st->valid && !st->rle) fse_release(&st->t);
    st->valid = 1; st->rle = 0;
    if (mode == 0) { set_default(&st->t, def, defn, deflog); return 0; }
    if (mode == 1) { st->rle = 1; st->t.nsym = codes[0]; bw_add(w, codes[0], 8); return 0; }
    uint32_t cnt[64] = {0}; int top = 0, distinct = 0;
    for (uint32_t i = 0; i < n; i++) { if (!cnt[codes[i]]) distinct++; cnt[codes[i]]++; if (codes[i] > top) top = codes[i]; }
    int nsym = top + 1;
    /* optionally pretend a few more symbols were seen once (gives "less than 1" cells and
       keeps single-symbol histograms FSE-describable, as a real encoder's dominant-symbol
       tables are) */
    int extra = sprinkle_low > 0 ? sprinkle_low : (distinct < 2 ? 1 : 0);
    for (int k = 0; k < extra; k++) { int s = (int)rng_below(r, (uint32_t)nsym_max); if (!cnt[s]) { cnt[s] = 1; if (s + 1 > nsym) nsym = s + 1; } }
    fse_normalize(&st->t, cnt, nsym, maxlog);
    if (sprinkle_low >= 0) for (int s = 0; s < nsym; s++) if (st->t.norm[s] == 1 && cnt[s] * (1u << maxlog) < n / 2 + 1) st->t.norm[s] = -1;
    fse_build(&st->t);
    fse_write_ncount(w, &st->t);
    return 0;
}

/* Sequences_Section for `n` sequences (n >= 1).  modes[3] = LL, OF, ML modes.
   Returns bytes written, 0 on failure. */
static size_t write_sequences_section(uint8_t* out, size_t cap, const seq_t* q, uint32_t n, const int modes[3],
                                      seq_table tabs[3], int sprinkle, rng_t* r) {
    uint8_t* llc = (uint8_t*)malloc(n), *mlc = (uint8_t*)malloc(n), *ofc = (uint8_t*)malloc(n);
    for (uint32_t i = 0; i < n; i++) { llc[i] = (uint8_t)ll_code(q[i].ll); mlc[i] = (uint8_t)ml_code(q[i].ml); ofc[i] = (uint8_t)highbit32(q[i].ofv); }
    bw_t w; bw_init(&w, out, cap);
    if (n < 128) bw_add(&w, n, 8);
    else if (n < 0x7F00) { bw_add(&w, (n >> 8) + 128, 8); bw_add(&w, n & 255, 8); }
    else { bw_add(&w, 255, 8); bw_add(&w, (n - 0x7F00) & 255, 8); bw_add(&w, (n - 0x7F00) >> 8, 8); }
    bw_add(&w, (uint64_t)((modes[0] << 6) | (modes[1] << 4) | (modes[2] << 2)), 8);
    int bad = 0;
    bad |= prepare_table(&tabs[0], modes[0], llc, n, 36, 9, LL_DEFAULT, 36, 6, &w, sprinkle, r);
    bad |= prepare_table(&tabs[1], modes[1], ofc, n, 29, 8, OF_DEFAULT, 29, 5, &w, sprinkle, r);
    bad |= prepare_table(&tabs[2], modes[2], mlc, n, 53, 9, ML_DEFAULT, 53, 6, &w, sprinkle, r);
    size_t hdr = bw_bytes(&w);
    if (bad || w.overflow) { free(llc); free(mlc); free(ofc); return 0; }
    /* every code must be encodable by its table */
    const fse_t* T[3] = { &tabs[0].t, &tabs[1].t, &tabs[2].t };
    const uint8_t* C[3] = { llc, ofc, mlc };
    for (int k = 0; k < 3 && !bad; k++) for (uint32_t i = 0; i < n; i++) {
        if (tabs[k].rle) { if (C[k][i] != T[k]->nsym) { bad = 1; break; } }
        else if (C[k][i] >= T[k]->nsym || T[k]->norm[C[k][i]] == 0) { bad = 1; break; }
    }
    if (bad) { free(llc); free(mlc); free(ofc); return 0; }
    bw_t s; bw_init(&s, out + hdr, cap - hdr);
    uint32_t sz[3] = { 1u << T[0]->log, 1u << T[1]->log, 1u << T[2]->log };
    uint32_t sLL = tabs[0].rle ? 0 : T[0]->first_state[llc[n - 1]];
    uint32_t sOF = tabs[1].rle ? 0 : T[1]->first_state[ofc[n - 1]];
    uint32_t sML = tabs[2].rle ? 0 : T[2]->first_state[mlc[n - 1]];
    for (uint32_t i = n; i-- > 0;) {
        if (i < n - 1) {                 /* transition i -> i+1, read order LL, ML, OF => written OF, ML, LL */
            if (!tabs[1].rle) { uint32_t p = T[1]->map[ofc[i] * sz[1] + sOF]; bw_add(&s, sOF - T[1]->base[p], T[1]->nb[p]); sOF = p; }
            if (!tabs[2].rle) { uint32_t p = T[2]->map[mlc[i] * sz[2] + sML]; bw_add(&s, sML - T[2]->base[p], T[2]->nb[p]); sML = p; }
            if (!tabs[0].rle) { uint32_t p = T[0]->map[llc[i] * sz[0] + sLL]; bw_add(&s, sLL - T[0]->base[p], T[0]->nb[p]); sLL = p; }
        }
        /* extra bits, read order OF, ML, LL => written LL, ML, OF */
        bw_add(&s, q[i].ll - LL_BASE[llc[i]], LL_BITS[llc[i]]);
        bw_add(&s, q[i].ml - ML_BASE[mlc[i]], ML_BITS[mlc[i]]);
        bw_add(&s, q[i].ofv - (1u << ofc[i]), ofc[i]);
    }
    /* initial states, read order LL, OF, ML => written ML, OF, LL */
    if (!tabs[2].rle) bw_add(&s, sML, (unsigned)T[2]->log);
    if (!tabs[1].rle) bw_add(&s, sOF, (unsigned)T[1]->log);
    if (!tabs[0].rle) bw_add(&s, sLL, (unsigned)T[0]->log);
    size_t body = bw_close_reversed(&s);
    free(llc); free(mlc); free(ofc);
    if (s.overflow) return 0;
    return hdr + body;
}

/* ------------------------------------------------------------ offsets */
typedef struct { uint32_t h[3]; } hist_t;
/* decoder-side resolution (sequence_execution.cairo:85-129); returns actual offset (0 = invalid) */
static uint32_t hist_apply(hist_t* H, uint32_t ofv, uint32_t ll) {
    uint32_t* h = H->h, a;
    if (ll > 0) a = ofv == 1 ? h[0] : ofv == 2 ? h[1] : ofv == 3 ? h[2] : ofv - 3;
    else a = ofv == 1 ? h[1] : ofv == 2 ? h[2] : ofv == 3 ? h[0] - 1 : ofv - 3;
    if (a == 0) return 0;
    if (ll > 0) { if (ofv == 1) {} else if (ofv == 2) { h[1] = h[0]; h[0] = a; } else { h[2] = h[1]; h[1] = h[0]; h[0] = a; } }
    else { if (ofv == 1) { h[1] = h[0]; h[0] = a; } else { h[2] = h[1]; h[1] = h[0]; h[0] = a; } }
    return a;
}
/* choose an offset_value valid at `produced` bytes (this sequence's literals included) */
static uint32_t pick_offset(rng_t* r, hist_t* H, uint32_t ll, uint64_t produced, uint32_t max_off, uint32_t rep_pct) {
    if (rng_below(r, 100) < rep_pct) {
        uint32_t v = 1 + rng_below(r, 3); hist_t t = *H; uint32_t a = hist_apply(&t, v, ll);
        if (a && a <= produced && a <= max_off) { *H = t; return v; }
    }
    uint64_t lim = produced < max_off ? produced : max_off;
    uint32_t a = 1 + rng_below(r, (uint32_t)lim);
    hist_apply(H, a + 3, ll);
    return a + 3;
}

/* ------------------------------------------------------ literal sources */
/* 64-symbol geometric-ish distribution (~5.2 bits/symbol) over a per-block random subset of
   byte values; `lowhalf` keeps values < 128 so the tree fits a direct description */
static void gen_literals(rng_t* r, uint8_t* lit, size_t n, int lowhalf, int nsyms) {
    uint8_t perm[256]; int range = lowhalf ? 128 : 256;
    for (int i = 0; i < range; i++) perm[i] = (uint8_t)i;
    for (int i = 0; i < nsyms; i++) { int j = i + (int)rng_below(r, (uint32_t)(range - i)); uint8_t t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    uint8_t lut[4096]; double ratio = 0.90 + 0.06 * (double)rng_below(r, 1000) / 1000.0, wsum = 0, w = 1;
    for (int k = 0; k < nsyms; k++) { wsum += w; w *= ratio; }
    int pos = 0; w = 1;
    for (int k = 0; k < nsyms; k++) {
        int cells = (int)(w / wsum * 4096.0 + 0.5); if (cells < 1) cells = 1; if (k == nsyms - 1) cells = 4096 - pos;
        for (int c = 0; c < cells && pos < 4096; c++) lut[pos++] = perm[k];
        w *= ratio;
    }
    while (pos < 4096) lut[pos++] = perm[0];
    size_t i = 0;
    while (i < n) { uint64_t x = rng_next(r); for (int k = 0; k < 5 && i < n; k++) { lit[i++] = lut[x & 4095]; x >>= 12; } }
}

/* ------------------------------------------------------------- frames */
typedef struct {
    uint8_t* out; size_t cap, len;       /* frame bytes */
    uint64_t regen;                       /* decompressed size of the frame */
    huf_t huf; int huf_valid;             /* carried Huffman table (Treeless) */
    seq_table tabs[3];
    hist_t hist;
    uint32_t window;
} frame_t;

static void put_block_header(uint8_t* p, int last, int type, uint32_t size) {
    uint32_t v = (uint32_t)last | ((uint32_t)type << 1) | (size << 3); p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16);
}
/* frame header: magic, descriptor (4-byte FCS, no checksum, no dict), window byte, FCS patched at the end */
static size_t frame_begin(frame_t* f, int window_log) {
    uint8_t* p = f->out; uint32_t magic = 0xFD2FB528u; memcpy(p, &magic, 4);
    p[4] = 0x80; p[5] = (uint8_t)((window_log - 10) << 3); memset(p + 6, 0, 4);
    f->len = 10; f->regen = 0; f->huf_valid = 0; f->hist.h[0] = 1; f->hist.h[1] = 4; f->hist.h[2] = 8;
    f->window = 1u << window_log;
    for (int k = 0; k < 3; k++) { f->tabs[k].valid = 0; f->tabs[k].rle = 0; f->tabs[k].t.map = NULL; }
    return 10;
}
static void frame_end(frame_t* f) {
    uint32_t fcs = (uint32_t)f->regen; memcpy(f->out + 6, &fcs, 4);
    for (int k = 0; k < 3; k++) if (f->tabs[k].valid && !f->tabs[k].rle) fse_release(&f->tabs[k].t);
}

typedef struct {
    int lit_type;        /* 0 raw 1 rle 2 huffman 3 treeless */
    int streams;         /* 1 or 4 */
    int tree_mode;       /* 0 direct 1 fse 2 auto */
    int lit_syms;        /* alphabet size of the literal source */
    uint32_t nlit;       /* literals in the block */
    uint32_t nseq;
    int modes[3];        /* LL, OF, ML */
    int ll_kind;         /* 0: pairs summing to 2 in {0,1,2}; 1: {0,1}; 2: wide random */
    int ml_kind;         /* 0: always 3; 2: wide random */
    uint32_t max_off; uint32_t rep_pct; int sprinkle;
} block_spec;

/* appends one compressed block; returns 0 on success */
static int add_compressed_block(frame_t* f, rng_t* r, const block_spec* sp, int last) {
    uint8_t* lit = (uint8_t*)malloc((size_t)sp->nlit + 16);
    seq_t* q = sp->nseq ? (seq_t*)malloc((size_t)sp->nseq * sizeof(seq_t)) : NULL;
    int rc = 1;
    /* literals */
    if (sp->lit_type == 1) memset(lit, (int)rng_below(r, 256), sp->nlit ? sp->nlit : 1);
    else if (sp->lit_type == 0) { for (uint32_t i = 0; i < sp->nlit; i += 8) { uint64_t x = rng_next(r); memcpy(lit + i, &x, 8); } }
    else if (sp->lit_type == 3 && f->huf_valid) {
        /* Treeless: draw only symbols the carried table can code */
        uint8_t ok[256]; int nok = 0; for (int s = 0; s < 256; s++) if (f->huf.len[s]) ok[nok++] = (uint8_t)s;
        for (uint32_t i = 0; i < sp->nlit; i++) { uint32_t a = rng_below(r, (uint32_t)nok), b = rng_below(r, (uint32_t)nok); lit[i] = ok[a < b ? a : b]; }
    } else gen_literals(r, lit, sp->nlit, sp->tree_mode == 0, sp->lit_syms);
    /* sequences */
    uint64_t produced = f->regen; uint32_t lit_used = 0; uint64_t out_bytes = 0;
    hist_t H = f->hist;
    for (uint32_t i = 0; i < sp->nseq; i++) {
        uint32_t ll, ml;
        if (sp->ll_kind == 0) { if ((i & 1) == 0) { ll = rng_below(r, 4); ll = ll == 3 ? 1 : ll; if (i == 0 && produced == 0 && ll == 0) ll = 1; } else ll = 2 - q[i - 1].ll; }
        else if (sp->ll_kind == 1) { ll = rng_below(r, 2); if (i == 0 && produced == 0) ll = 1; }
        else { uint32_t k = rng_below(r, 100); ll = k < 60 ? rng_below(r, 8) : k < 95 ? rng_below(r, 64) : rng_below(r, 3000); if (i == 0 && produced == 0 && ll == 0) ll = 1; }
        if (lit_used + ll > sp->nlit) ll = sp->nlit - lit_used;
        if (produced + ll == 0) { goto done; }
        if (sp->ml_kind == 0) ml = 3;
        else { uint32_t k = rng_below(r, 1000); ml = k < 700 ? 3 + rng_below(r, 16) : k < 970 ? 3 + rng_below(r, 200) : k < 998 ? 3 + rng_below(r, 3000) : 3 + rng_below(r, 70000); }
        lit_used += ll; produced += ll;
        q[i].ll = ll; q[i].ml = ml; q[i].ofv = pick_offset(r, &H, ll, produced, sp->max_off < f->window ? sp->max_off : f->window, sp->rep_pct);
        produced += ml; out_bytes += ll + ml;
    }
    out_bytes += sp->nlit - lit_used;
    {
        uint8_t* blk = f->out + f->len + 3; size_t room = f->cap - f->len - 3, n1, n2 = 0;
        if (room > 131072 + 2048) room = 131072 + 2048;
        huf_t h; const huf_t* hp = &f->huf;
        int lt = sp->lit_type;
        if (lt == 3 && !f->huf_valid) lt = 2;
        if (lt == 2) {
            uint32_t cnt[256] = {0}; for (uint32_t i = 0; i < sp->nlit; i++) cnt[lit[i]]++;
            if (huf_build(&h, cnt)) goto done;
            hp = &h;
        }
        n1 = write_literals_section(blk, room, lt, lit, sp->nlit, hp, sp->streams, sp->tree_mode);
        if (!n1) goto done;
        if (sp->nseq == 0) { blk[n1] = 0; n2 = 1; }
        else {
            n2 = write_sequences_section(blk + n1, room - n1, q, sp->nseq, sp->modes, f->tabs, sp->sprinkle, r);
            if (!n2 || n1 + n2 > 128 * 1024) {
                /* the block is dropped: the decoder never sees these tables, so forget them */
                for (int k = 0; k < 3; k++) { if (f->tabs[k].valid && !f->tabs[k].rle) fse_release(&f->tabs[k].t); f->tabs[k].valid = 0; f->tabs[k].rle = 0; }
                goto done;
            }
        }
        if (n1 + n2 > 128 * 1024) goto done;
        if (lt == 2) { f->huf = h; f->huf_valid = 1; }
        put_block_header(f->out + f->len, last, 2, (uint32_t)(n1 + n2));
        f->len += 3 + n1 + n2; f->regen += out_bytes; f->hist = H;
        rc = 0;
    }
done:
    free(lit); free(q); return rc;
}
static void add_raw_block(frame_t* f, rng_t* r, uint32_t n, int last) {
    put_block_header(f->out + f->len, last, 0, n);
    uint8_t* p = f->out + f->len + 3;
    for (uint32_t i = 0; i < n; i += 8) { uint64_t x = rng_next(r); memcpy(p + i, &x, (n - i) >= 8 ? 8 : (n - i)); }
    f->len += 3 + n; f->regen += n;
}
static void add_rle_block(frame_t* f, uint8_t byte, uint32_t n, int last) {
    put_block_header(f->out + f->len, last, 1, n); f->out[f->len + 3] = byte; f->len += 4; f->regen += n;
}

static int gen_frame(frame_t* f, int kind, uint64_t seed, uint64_t index) {
    rng_t r = { seed ^ (index * 0x9E3779B97F4A7C15ULL) ^ ((uint64_t)kind << 56) }; rng_next(&r);
    block_spec sp; memset(&sp, 0, sizeof sp);
    for (int attempt = 0; attempt < 8; attempt++) {
        frame_begin(f, 17);
        int rc = 0;
        switch (kind) {
        case CZS_RAW_RLE:                         /* config 2: 50 % Raw 131072 random / 50 % RLE, byte = index & 0xFF */
            if (index & 1) add_rle_block(f, (uint8_t)(index & 0xFF), 131072, 1); else add_raw_block(f, &r, 131072, 1);
            break;
        case CZS_HUF_LITERALS:                    /* config 3 */
            sp.lit_type = 2; sp.streams = 4; sp.tree_mode = (index & 1) ? 0 : 1; sp.lit_syms = 64; sp.nlit = 131072; sp.nseq = 0;
            rc = add_compressed_block(f, &r, &sp, 1); break;
        case CZS_FULL_4A:                         /* config 4a */
            sp.lit_type = 2; sp.streams = 4; sp.tree_mode = 2; sp.lit_syms = 64; sp.nlit = 32768; sp.nseq = 32768;
            sp.modes[0] = sp.modes[1] = sp.modes[2] = 2; sp.ll_kind = 0; sp.ml_kind = 0; sp.max_off = 131072; sp.rep_pct = 25; sp.sprinkle = 3;
            rc = add_compressed_block(f, &r, &sp, 1); break;
        case CZS_FULL_4B:                         /* config 4b: 65536 sequences, ll in {0,1}, ml = 3 (out ~224 KiB) */
            sp.lit_type = 2; sp.streams = 4; sp.tree_mode = 2; sp.lit_syms = 64; sp.nlit = 33000; sp.nseq = 65536;
            sp.modes[0] = sp.modes[1] = sp.modes[2] = 2; sp.ll_kind = 1; sp.ml_kind = 0; sp.max_off = 4096; sp.rep_pct = 50; sp.sprinkle = 3;
            rc = add_compressed_block(f, &r, &sp, 1); break;
        default: {                                /* config 5: corpus-like mix, multi-block frames */
            frame_begin(f, 17 + (int)rng_below(&r, 4));
            uint32_t k = rng_below(&r, 100), nblocks = k < 55 ? 1 : k < 80 ? 2 + rng_below(&r, 3) : k < 97 ? 5 + rng_below(&r, 12) : 20 + rng_below(&r, 30);
            for (uint32_t b = 0; b < nblocks && !rc; b++) {
                int last = b == nblocks - 1; uint32_t t = rng_below(&r, 100);
                if (f->len + 140000 > f->cap) { last = 1; t = 99; }
                if (t < 20) { uint32_t n = rng_below(&r, 100) < 80 ? rng_below(&r, 2000) : rng_below(&r, 131073); if (f->len + n + 1024 > f->cap) n = 16; add_raw_block(f, &r, n, last); }
                else if (t < 32) add_rle_block(f, (uint8_t)rng_next(&r), rng_below(&r, 100) < 70 ? rng_below(&r, 5000) : rng_below(&r, 131073), last);
                else {
                    memset(&sp, 0, sizeof sp);
                    uint32_t szk = rng_below(&r, 100);
                    sp.nlit = szk < 50 ? rng_below(&r, 600) : szk < 90 ? rng_below(&r, 8000) : rng_below(&r, 70000);
                    uint32_t lt = rng_below(&r, 100);
                    sp.lit_type = lt < 6 ? 0 : lt < 13 ? 1 : lt < 78 ? 2 : 3;
                    sp.lit_syms = 2 + (int)rng_below(&r, 120);
                    if (sp.lit_type >= 2 && sp.nlit < 8) sp.lit_type = 0;
                    if (sp.lit_type == 1 && sp.nlit == 0) sp.lit_type = 0;
                    sp.streams = (sp.nlit >= 1024 || rng_below(&r, 100) < 40) ? 4 : 1;
                    if (sp.streams == 4 && sp.nlit < 64) sp.streams = 1;
                    sp.tree_mode = (int)rng_below(&r, 3);
                    if (sp.lit_type == 0 && sp.nlit > 60000) sp.nlit = 60000;
                    uint32_t sq = rng_below(&r, 100);
                    sp.nseq = sq < 5 ? 0 : sq < 60 ? 1 + rng_below(&r, 60) : sq < 92 ? rng_below(&r, 1500) : rng_below(&r, 20000);
                    if (sp.nlit == 0 && f->regen == 0) sp.nseq = 0;
                    for (int m = 0; m < 3; m++) { uint32_t mk = rng_below(&r, 100); sp.modes[m] = mk < 25 ? 0 : mk < 29 ? 1 : mk < 92 ? 2 : 3; }
                    sp.ll_kind = 2; sp.ml_kind = 2; sp.max_off = 1u << (10 + rng_below(&r, 11)); sp.rep_pct = rng_below(&r, 60); sp.sprinkle = (int)rng_below(&r, 4);
                    if (sp.nseq > 3000) { sp.ml_kind = 0; }   /* keep regenerated size bounded */
                    /* RLE table modes need constant codes: fall back to FSE for those */
                    int ok = 1;
                    for (int tries = 0; tries < 4; tries++) {
                        block_spec s2 = sp; ok = !add_compressed_block(f, &r, &s2, last);
                        if (ok) break;
                        for (int m = 0; m < 3; m++) if (sp.modes[m] == 1 || sp.modes[m] == 3 || sp.modes[m] == 0) sp.modes[m] = 2;
                        if (tries >= 1) { sp.lit_type = sp.lit_type == 3 ? 2 : sp.lit_type; sp.tree_mode = 2; }
                        if (tries >= 2) { sp.nseq = sp.nseq > 200 ? 200 : sp.nseq; sp.nlit = sp.nlit > 30000 ? 30000 : sp.nlit; }
                    }
                    if (!ok) add_rle_block(f, 0x5A, 100, last);
                }
            }
        } }
        if (!rc) { frame_end(f); return 0; }
        frame_end(f);
    }
    /* could not build the requested shape: emit an empty raw frame so the batch stays valid */
    frame_begin(f, 17); add_raw_block(f, &r, 0, 1); frame_end(f);
    return 1;
}

/* --------------------------------------------------------------- API */
typedef struct {
    int kind; uint64_t seed, first_index; size_t n, stride; uint8_t* base; uint64_t* len; uint64_t* regen;
    volatile size_t* next; volatile int* fails;
} gen_job;
static void* gen_worker(void* arg) {
    gen_job* j = (gen_job*)arg;
    for (;;) {
        size_t i = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (i >= j->n) break;
        frame_t f; memset(&f, 0, sizeof f);
        f.out = j->base + i * j->stride; f.cap = j->stride;
        if (gen_frame(&f, j->kind, j->seed, j->first_index + i)) __atomic_fetch_add(j->fails, 1, __ATOMIC_RELAXED);
        j->len[i] = f.len; j->regen[i] = f.regen;
    }
    return NULL;
}
/* Bytes to reserve per frame slot for `kind`. */
CZS_API size_t czs_slot_bytes(int kind) {
    switch (kind) { case CZS_RAW_RLE: return 131072 + 64; case CZS_MIX: return 1u << 20; default: return 131072 + 4096; }
}
/* Generates frames first_index .. first_index+n-1 of workload `kind` into fixed-stride slots
   base[i*stride ..].  len[i] = frame bytes, regen[i] = decompressed bytes.  Returns the number
   of frames that fell back to the empty placeholder (0 expected). */
CZS_API int czs_generate(int kind, uint64_t seed, uint64_t first_index, size_t n, uint8_t* base, size_t stride,
                         uint64_t* len, uint64_t* regen, int nthreads) {
    volatile size_t next = 0; volatile int fails = 0;
    gen_job j = { kind, seed, first_index, n, stride, base, len, regen, &next, &fails };
    if (nthreads <= 1) { gen_worker(&j); return fails; }
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, gen_worker, &j);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    return fails;
}
