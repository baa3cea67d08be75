/*
 * TEST INFRASTRUCTURE ONLY — runs cz_decode_frames_kernel (the unmodified kernel source) on
 * the CPU through tests/emu/hip/hip_runtime.h, normally under ASan+UBSan.
 * usage: emu_decode <batch.bin> <result.bin>
 *   batch.bin : u64 n, then n x { u64 in_len, u64 out_cap, in bytes }
 *   result.bin: n x { cz_frame_result, out bytes (bytes_produced) }
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

thread_local emu_dim3 threadIdx;
thread_local emu_dim3 blockIdx;
emu_dim3 gridDim;
emu_dim3 blockDim;
pthread_barrier_t emu_barrier;
pthread_barrier_t emu_wbar[EMU_MAX_WAVES];
volatile uint64_t emu_xchg_all[EMU_MAX_WAVES][64];
void* volatile emu_site[EMU_MAX_THREADS];
void* volatile emu_ring[EMU_MAX_THREADS][64];
volatile uint64_t emu_sync_count[EMU_MAX_THREADS];
static volatile int emu_lane_done[EMU_MAX_THREADS];
static volatile int emu_nthreads = 64;
#include <execinfo.h>
#include <unistd.h>
/* watchdog: if no lane passes a barrier for 20 s, print where every lane waits and abort */
static void* emu_watchdog(void*) {
    uint64_t last = 0; int idle = 0;
    for (;;) {
        sleep(1);
        uint64_t sum = 0; for (int i = 0; i < EMU_MAX_THREADS; i++) sum += emu_sync_count[i];
        if (sum != last) { last = sum; idle = 0; continue; }
        if (++idle < 20) continue;
        fprintf(stderr, "EMU HANG: barrier sites per lane (addr2line -e emu_decode <addr>):\n");
        for (int i = 0; i < emu_nthreads; i++) fprintf(stderr, "lane %d done=%d syncs=%llu site=%p\n", i, emu_lane_done[i], (unsigned long long)emu_sync_count[i], emu_site[i]);
        for (int l = 0; l < 2; l++) { fprintf(stderr, "ring lane %d:", l); for (int k = 0; k < 64; k++) fprintf(stderr, " %p", emu_ring[l][(emu_sync_count[l] + 1 + k) & 63]); fprintf(stderr, "\n"); }
        _exit(3);
    }
    return nullptr;
}

#include "czstd_kernels.hip"
#include "czstd_chain.hip"
#include "czstd_pre.hip"
#include "czstd_wexec.hip"
#define CZ_EXEC_ONLY 1
namespace czx {
#include "czstd_kernels.hip"
}
#undef CZ_EXEC_ONLY

struct lane_arg { cz_batch_args a; unsigned lane, block; int which; const uint8_t* dict_raw; uint64_t dict_len; cz_device_frame_state* dict_state; uint64_t* dict_res; };
static void* lane_main(void* p) {
    lane_arg* la = (lane_arg*)p;
    threadIdx.x = la->lane; blockIdx.x = la->block;
    emu_lane_done[la->lane] = 0;
    if (la->which == 0 || la->which == 12) cz_chain_kernel(la->a); else if (la->which == 2 || la->which == 11 || la->which == 13) czx::cz_execute_frames_kernel(la->a);
    else if (la->which == 14) cz_wexec_kernel(la->a);
    else if (la->which == 6) cz_dict_setup_kernel(la->dict_raw, la->dict_len, la->dict_state, la->dict_res);
    else if (la->which == 7) cz_huf_kernel(la->a);
    else if (la->which == 9) cz_huf1_kernel(la->a);
    else if (la->which == 8) cz_tile_kernel(la->a);
    else if (la->which == 10) cz_wexec_kernel(la->a);
    else if (la->which >= 4) cz_scan_kernel(la->a);                     /* 4, 5: the two passes of the block scan */
    else cz_decode_frames_kernel(la->a);
    emu_lane_done[la->lane] = 1;
    return nullptr;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
    uint64_t n; if (fread(&n, 8, 1, f) != 1) return 2;
    std::vector<uint64_t> in_off(n), in_len(n), out_off(n), out_cap(n);
    std::vector<uint8_t> in; uint64_t out_total = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t l, c; if (fread(&l, 8, 1, f) != 1 || fread(&c, 8, 1, f) != 1) return 2;
        in_off[i] = in.size(); in_len[i] = l; out_cap[i] = c; out_off[i] = out_total; out_total += c;
        size_t at = in.size(); in.resize(at + l);
        if (l && fread(in.data() + at, 1, l, f) != l) return 2;
    }
    fclose(f);
    /* exact-size heap blocks so that ASan sees any byte read or written out of range */
    uint8_t* in_exact = (uint8_t*)malloc(in.size() ? in.size() : 1); if (in.size()) memcpy(in_exact, in.data(), in.size());
    uint8_t* out = (uint8_t*)malloc(out_total ? out_total : 1); memset(out, 0xEE, out_total);
    std::vector<cz_frame_result> res(n);
    uint32_t counter = 0;
    const int grid = 2;
    uint8_t* lit = (uint8_t*)malloc((size_t)grid * CZ_WG_SCRATCH_BYTES);
    cz_batch_args a; memset(&a, 0, sizeof a);
    a.in_base = in_exact; a.in_off = in_off.data(); a.in_len = in_len.data();
    a.out_base = out; a.out_off = out_off.data(); a.out_cap = out_cap.data();
    a.results = res.data(); a.tasks = nullptr; a.n = (uint32_t)n; a.work_counter = &counter;
    a.lit_scratch = lit; a.lit_scratch_stride = CZ_WG_SCRATCH_BYTES; a.verify_checksum = getenv("EMU_VERIFY") ? (uint32_t)atoi(getenv("EMU_VERIFY")) : 1u;
    /* EMU_CHAIN=<bytes>: run the FSE-chain pre-pass first, with an arena of that many bytes */
    const char* ce = getenv("EMU_CHAIN");
    unsigned long long chain_top[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint32_t chain_counter = 0;
    std::vector<uint64_t> frame_first(n ? n : 1, 0);
    uint64_t* arena = nullptr;
    if (ce && atoll(ce) > 0) {
        size_t bytes = (size_t)atoll(ce);
        arena = (uint64_t*)malloc(bytes); a.chain_arena = arena; a.chain_capacity = bytes / 8; a.chain_top = chain_top;
        a.frame_first = frame_first.data(); a.chain_counter = &chain_counter;
        a.chain_min_nseq = getenv("EMU_CHAIN_MIN") ? (uint32_t)atoi(getenv("EMU_CHAIN_MIN")) : 0;
    }
    { pthread_t wd; pthread_create(&wd, nullptr, emu_watchdog, nullptr); pthread_detach(wd); }
    for (int w = 0; w < EMU_MAX_WAVES; w++) pthread_barrier_init(&emu_wbar[w], nullptr, 64);
    uint32_t exec_counter = 0; a.exec_counter = &exec_counter; a.exec_variant_force = 4;   /* (the emulator has one build of cz_execute_frames_kernel) */
    /* EMU_LIT=<bytes>: the literal / copy half of the pre-pass (cz_huf_kernel, cz_tile_kernel) with a literal arena of that many bytes */
    unsigned long long lit_top[4] = {0, 0, 0, 0}; std::vector<uint64_t> lit_first(n ? n : 1, 0); uint8_t* lit_arena = nullptr;
    const size_t lit_bytes = arena && getenv("EMU_LIT") ? (size_t)atoll(getenv("EMU_LIT")) : 0;
    std::vector<cz_lit_seg> lit_segs; std::vector<cz_copy_seg> copy_segs; std::vector<uint32_t> frame_pre(n ? n : 1, 0);
    if (lit_bytes) {
        lit_arena = (uint8_t*)malloc(lit_bytes); a.lit_arena = lit_arena; a.lit_capacity = lit_bytes; a.lit_top = lit_top; a.lit_first = lit_first.data();
        const size_t cap = getenv("EMU_SEGS") ? (size_t)atoll(getenv("EMU_SEGS")) : lit_bytes / 256 + 4096;
        lit_segs.resize(cap); copy_segs.resize(cap);
        a.lit_segs = lit_segs.data(); a.lit_seg_capacity = (uint32_t)cap; a.copy_segs = copy_segs.data(); a.copy_seg_capacity = (uint32_t)cap; a.frame_pre = frame_pre.data();
    }
    /* launches: [scan, huff0 and tile kernels, chain pre-pass, [cz_execute_frames_kernel (EMU_EXEC=1),]] main kernel (behind cz_execute_frames_kernel: the frames it left) */
    const int with_exec = arena && lit_bytes && getenv("EMU_EXEC") && atoi(getenv("EMU_EXEC")) > 0;
    /* EMU_WEXEC=<waves>: cz_wexec_kernel (that many waves per workgroup) ahead of cz_execute_frames_kernel */
    const int wx_waves = with_exec && getenv("EMU_WEXEC") ? atoi(getenv("EMU_WEXEC")) : 0;
    std::vector<uint32_t> wx_list(n ? n : 1, 0); uint32_t wx_counter = 0;
    a.debug_flags = getenv("EMU_DEBUG_FLAGS") ? (uint32_t)atoi(getenv("EMU_DEBUG_FLAGS")) : 0u;   /* CZ_DEBUG_* */
    if (wx_waves > 0) { a.wx_list = wx_list.data(); a.wx_counter = &wx_counter; a.wx_force = getenv("EMU_WX_AUTO") ? 0u : 1u; a.wx_leave = 0; }   /* forced on unless EMU_WX_AUTO: the batch's offset codes decide, as on the device */
    uint32_t fallback_count = 0; std::vector<uint32_t> fallback_list(n ? n : 1, 0);
    std::vector<cz_blk_desc> blk_desc; std::vector<uint32_t> scan_ctl(CZ_SCAN_CTL_WORDS, 0), frame_order, scan_wave;
    if (arena) {
        blk_desc.resize(a.chain_capacity / (4 + CZ_CHAIN_MAP_WORDS + 1) + 4096);
        a.blk_desc = blk_desc.data(); a.blk_capacity = (uint32_t)blk_desc.size(); a.scan_ctl = scan_ctl.data();
        frame_order.resize(n ? n : 1); a.frame_order = frame_order.data();
        scan_wave.assign(((n + 63) / 64 + 1) * 72, 0); a.scan_wave = scan_wave.data();
    }
    /* EMU_DICT=<file>: parse the dictionary with cz_dict_setup_kernel (one workgroup) and start every frame from it, as
       cz_context_set_dictionary does; a dictionary that does not parse ends the run with exit code 3 and its status on stderr */
    std::vector<uint8_t> dict_raw; cz_device_frame_state* dict_state = nullptr; uint64_t dict_res[4] = {0, 0, 0, 0};
    if (const char* de = getenv("EMU_DICT")) {
        FILE* df = fopen(de, "rb"); if (!df) return 2;
        fseek(df, 0, SEEK_END); long dl = ftell(df); fseek(df, 0, SEEK_SET);
        dict_raw.resize((size_t)dl); if (dl && fread(dict_raw.data(), 1, (size_t)dl, df) != (size_t)dl) return 2;
        fclose(df);
        uint8_t* dict_exact = (uint8_t*)malloc(dict_raw.size() ? dict_raw.size() : 1); memcpy(dict_exact, dict_raw.data(), dict_raw.size());
        dict_state = (cz_device_frame_state*)calloc(1, sizeof(cz_device_frame_state));
        emu_nthreads = 64; pthread_barrier_init(&emu_barrier, nullptr, 64u);
        std::vector<pthread_t> th(64); std::vector<lane_arg> la(64);
        for (int l = 0; l < 64; l++) {
            la[l].a = a; la[l].lane = (unsigned)l; la[l].block = 0; la[l].which = 6;
            la[l].dict_raw = dict_exact; la[l].dict_len = dict_raw.size(); la[l].dict_state = dict_state; la[l].dict_res = dict_res;
            pthread_create(&th[l], nullptr, lane_main, &la[l]);
        }
        for (int l = 0; l < 64; l++) pthread_join(th[l], nullptr);
        pthread_barrier_destroy(&emu_barrier);
        fprintf(stderr, "EMU_DICT: status %llu content offset %llu id %llu\n", (unsigned long long)dict_res[0], (unsigned long long)dict_res[1], (unsigned long long)dict_res[2]);
        if (dict_res[0]) return 3;
        a.dict_state = dict_state; a.dict = dict_exact + dict_res[1]; a.dict_len = dict_raw.size() - dict_res[1];
    }
    /* launches: [block scan (count, place), [cz_huf_kernel / cz_huf1_kernel, cz_tile_kernel (EMU_LIT),] chain pre-pass, [cz_execute_frames_kernel (EMU_EXEC=1),]] main kernel */
    if (with_exec) { a.fallback_list = fallback_list.data(); a.fallback_count = &fallback_count; }   /* as the host library: set before cz_huf_kernel, which may list frames too */
    /* EMU_HUF1=1: cz_huf1_kernel (the one-wave form that runs beside the chain kernel on the device) never sees the chain kernel
       "done" and takes every literals section; default: it sees it done at once and cz_huf_kernel takes them all */
    const int huf1_all = getenv("EMU_HUF1") && atoi(getenv("EMU_HUF1")) > 0;
    a.chain_grid = (uint32_t)grid;
    /* EMU_SPLIT=1: the chain pre-pass as two launches (0: the large blocks, published one by one; 12: the others) and the early launches
       of cz_execute_frames_kernel (13) and cz_wexec_kernel (14) ahead of the later ones, as cz_context_set_early_execute(1) arranges them
       on the device.  EMU_SPLIT=2: cz_wexec_kernel's early launch runs BEFORE the large blocks' chains, so every large block's flag
       stays 0 until the bound of its wait: the frame goes to cz_decode_frames_kernel. */
    const int emu_split = getenv("EMU_SPLIT") ? atoi(getenv("EMU_SPLIT")) : 0;
    uint32_t early_exec_counter = 0, early_wx_counter = 0;
    int order[14] = {4, 5, 0, 9, 7, 8, 10, 2, 11, 1, -1, -1, -1, -1};   /* 2 / 11: cz_execute_frames_kernel beside / behind cz_wexec_kernel (10) */
    if (emu_split == 1) { const int o[14] = {4, 5, 0, 12, 9, 7, 8, 13, 14, 10, 2, 1, -1, -1}; for (int i = 0; i < 14; i++) order[i] = o[i]; }
    if (emu_split == 2) { const int o[14] = {4, 5, 12, 9, 7, 8, 14, 0, 13, 10, 2, 1, -1, -1}; for (int i = 0; i < 14; i++) order[i] = o[i]; }
    /* EMU_EXEC_FIRST=1: cz_execute_frames_kernel ahead of cz_wexec_kernel — on the device the two run side by side, and which of them
       meets a frame first depends on timing; the emulator runs them one after the other, in either order */
    if (getenv("EMU_EXEC_FIRST") && atoi(getenv("EMU_EXEC_FIRST")) > 0) { order[6] = 2; order[7] = 10; }
    for (int pi = 0; pi < 14; pi++) {
        const int which = order[pi];
        if (which < 0 || (!arena && which != 1)) continue;
        if ((which == 2 || which == 11 || which == 13) && !with_exec) continue;
        if ((which == 10 || which == 14) && wx_waves <= 0) continue;
        if (which == 11) continue;
        if ((which == 7 || which == 8 || which == 9) && !lit_bytes) continue;
        const int nthreads = which == 7 ? CZH_THREADS : (which == 8 ? 256 : (which == 10 || which == 14 ? 64 * wx_waves : 64));
        blockDim.x = (unsigned)nthreads;

        const int nblocks = which == 4 || which == 5 ? (int)((n + 63) / 64) : (which == 7 || which == 8 || which == 9 || which == 10 || which == 14 ? 1 : grid);
        emu_nthreads = nthreads; gridDim.x = (unsigned)nblocks;
        pthread_barrier_init(&emu_barrier, nullptr, (unsigned)nthreads);
        for (int b = 0; b < nblocks; b++) {
            std::vector<pthread_t> th((size_t)nthreads); std::vector<lane_arg> la((size_t)nthreads);
            for (int l = 0; l < nthreads; l++) {
                la[l].a = a; la[l].lane = (unsigned)l; la[l].block = (unsigned)b; la[l].which = which;
                if (which == 4 || which == 5) la[l].a.scan_pass = (uint32_t)(which - 4);
                if (which == 9 && huf1_all) la[l].a.chain_grid = 0x7FFFFFFFu;
                if (emu_split) {
                    if (which == 0) la[l].a.chain_part = 1u;
                    if (which == 12) la[l].a.chain_part = 2u;
                    if (which == 13) { la[l].a.early = 1u; la[l].a.exec_counter = &early_exec_counter; }
                    if (which == 14) { la[l].a.early = 1u; la[l].a.wx_counter = &early_wx_counter; }
                    if (which == 10 || which == 2) la[l].a.early = 2u;
                }
                pthread_create(&th[l], nullptr, lane_main, &la[l]);
            }
            for (int l = 0; l < nthreads; l++) pthread_join(th[l], nullptr);
        }
        pthread_barrier_destroy(&emu_barrier);
    }
    if (lit_bytes) {
        unsigned long long nl = 0, np = 0; for (uint64_t i = 0; i < n; i++) { nl += lit_first[i] != 0; np += (frame_pre[i] & CZ_PRE_COUNT) != 0; }
        fprintf(stderr, "EMU_LIT: %llu frames have their literals done, %llu have leading blocks in place, arena top %llu, %u sections, %u runs\n", nl, np, lit_top[0],
                scan_ctl[168] + scan_ctl[169] + scan_ctl[170] + scan_ctl[171] + scan_ctl[172] + scan_ctl[173] + scan_ctl[174] + scan_ctl[175] + scan_ctl[176] + scan_ctl[177] + scan_ctl[178] + scan_ctl[179] + scan_ctl[180] + scan_ctl[181] + scan_ctl[182] + scan_ctl[183] + scan_ctl[184] + scan_ctl[185] + scan_ctl[186] + scan_ctl[187], scan_ctl[202]);
        free(lit_arena);
    }
    if (with_exec) {
        fprintf(stderr, "EMU_EXEC: %llu frames finished by cz_execute_frames_kernel\n", (unsigned long long)(n - fallback_count));
        /* a frame is handed to cz_decode_frames_kernel ONCE (cz_list_fallback), whoever hands it back */
        std::vector<int> seen(n ? n : 1, 0);
        for (uint32_t i = 0; i < fallback_count && i < n; i++) { if (fallback_list[i] >= n || seen[fallback_list[i]]++) { fprintf(stderr, "EMU_EXEC: frame %u is on the fall-back list twice (or not a frame)\n", fallback_list[i]); return 4; } }
        if (fallback_count > n) { fprintf(stderr, "EMU_EXEC: %u entries on the fall-back list of %llu frames\n", fallback_count, (unsigned long long)n); return 4; }
    }
    if (wx_waves > 0) { unsigned long long nd = 0; for (uint64_t i = 0; i < n; i++) nd += (frame_pre[i] & CZ_PRE_WXDONE) != 0; fprintf(stderr, "EMU_WEXEC: %u frames listed, %llu finished by cz_wexec_kernel\n", scan_ctl[206], nd); }
    FILE* g = fopen(argv[2], "wb"); if (!g) return 2;
    for (uint64_t i = 0; i < n; i++) {
        fwrite(&res[i], sizeof(cz_frame_result), 1, g);
        uint64_t w = res[i].bytes_produced <= out_cap[i] ? res[i].bytes_produced : out_cap[i];
        fwrite(out + out_off[i], 1, w, g);
    }
    fclose(g);
    if (arena) { unsigned long long used = 0; for (uint64_t i = 0; i < n; i++) used += frame_first[i] != 0; fprintf(stderr, "EMU_CHAIN: %llu of %llu frames have chain records, arena top %llu\n", used, (unsigned long long)n, chain_top[0]); }
    free(in_exact); free(out); free(lit); free(arena);
    return 0;
}
