/*
 * czstd_host.hip — host side of libcairo_zstd_amd.so: the C ABI of include/cairo_zstd_amd.h.
 *
 *   cz_context_*          device context: stream, per-workgroup literal scratch, work counter
 *   cz_decode_batch_*     batch planner + launch of cz_decode_frames_kernel
 *   cz_frame_decoder_*    C++ mirror of the reference's FrameDecoder state machine
 *                         (src/frame_decoder.cairo:107-335); block decoding itself is always
 *                         done by the device kernel — there is no CPU decode path here.
 *   cz_read_frame_header / cz_read_block_header   stateless parsers (frame.cairo:152-284,
 *                         block_decoder.cairo:237-278)
 */
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <new>
#include <queue>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "czstd_types.h"

#include "czstd_kernels.hip"   /* single translation unit: kernels + host side */
#include "czstd_chain.hip"
#include "czstd_pre.hip"
#include "czstd_wexec.hip"
#ifdef CZ_EXP_PAD   /* diagnostic: shifts the code objects behind it by CZ_EXP_PAD x 256 bytes (does the layout of the kernels in the code object matter?) */
extern "C" __global__ void cz_pad_kernel(uint32_t* p) {
#pragma unroll
    for (int i = 0; i < CZ_EXP_PAD * 32; i++) asm volatile("s_nop 0\n s_nop 0");
    if (p) p[0] = 1;
}
#endif
/* the same kernel source once more, without its decoders: cz_execute_frames_kernel (czstd_kernels.hip, CZ_EXEC_ONLY) */
#define CZ_EXEC_ONLY 1
namespace czx {
#include "czstd_kernels.hip"
}
#undef CZ_EXEC_KERNEL
/* ... and a third time with a register budget of 8 waves per SIMD: frames of short sequences with near offsets (config 4b), where
   the execute kernel waits on its own recent stores rather than on HBM, run a quarter faster with twice the waves; long sequences
   (the general loop) would spill at 64 registers.  Which of the two builds runs a batch is decided on the device (cz_exec_variant). */
#undef CZ_EXEC_WAVES
#define CZ_EXEC_WAVES 8
#define CZ_EXEC_KERNEL cz_execute_frames8_kernel
namespace czx8 {
#include "czstd_kernels.hip"
}
#undef CZ_EXEC_KERNEL
#undef CZ_EXEC_WAVES
#define CZ_EXEC_WAVES 4
#undef CZ_EXEC_ONLY

#define CZ_EXPORT extern "C" __attribute__((visibility("default")))
#define CZ_CTL_BLOCK_BYTES (192 + CZ_SCAN_CTL_WORDS * 4)
#ifndef CZ_WX_SPARE_WGS
#define CZ_WX_SPARE_WGS 0       /* workgroups of cz_wexec_kernel beyond those that stay */
#endif

/* ------------------------------------------------------------------ context */
/* Dictionary (src/decoding/dictionary.cairo:11-18): the raw bytes and the carried-state image decode_dict makes of them, both
 * in HBM.  The tables are built by cz_dict_setup_kernel with the decoder's own builders. */
struct cz_dictionary {
    struct cz_context* ctx = nullptr;
    uint8_t* d_raw = nullptr; size_t len = 0; size_t content_off = 0;
    cz_device_frame_state* d_state = nullptr;
    uint32_t id = 0; uint32_t hist[3] = {0, 0, 0};
};
struct cz_context {
    int device = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    int num_cu = 0, occupancy = 0, grid_max = 0;
    uint8_t* lit_scratch = nullptr; int lit_slots = 0; uint32_t* work_counter = nullptr;
    const struct cz_dictionary* batch_dict = nullptr;                   /* cz_context_set_dictionary */
    hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_mid2 = nullptr, ev_stop = nullptr; bool timed = false, timed_chain = false, timed_exec = false;
    bool wexec_kernel = true;              /* far-offset batches: cz_wexec_kernel (a workgroup per frame, window in LDS) side by side with cz_execute_frames_kernel */
    int wexec_cus = 0;                     /* CUs (= workgroups) cz_wexec_kernel runs on; 0: half of them */
    uint32_t debug_flags = 0;              /* CZ_DEBUG_* */
    uint32_t exec_variant_force = 0;       /* 0: cz_exec_variant decides; 4 / 8: that variant of cz_execute_frames_kernel (A/B runs) */
    uint32_t wexec_force = 0;              /* 0: the kernels decide from the batch's offset codes; 1: always side by side (A/B runs) */
    int wexec_leave_per_cu = 7;           /* frames per workgroup of cz_wexec_kernel that cz_execute_frames_kernel leaves to it at the end of a batch */
    bool wexec_ready = false; uint32_t* wx_list = nullptr; hipEvent_t ev_wx = nullptr; bool timed_wx = false;
    bool early_execute = false;            /* (off by default: measured, profiles/r5/NOTES.md) the chain pre-pass as two launches (large blocks / all others) and the early execute launches behind the second (cz_launch) */
    hipStream_t stream4 = nullptr; hipEvent_t ev_small = nullptr, ev_e1 = nullptr, ev_w1 = nullptr, ev_x4 = nullptr; bool timed_small = false;
    bool exec_kernel = true;               /* frames the pre-pass finished (chain records + literals) run on cz_execute_frames_kernel; 0: all on cz_decode_frames_kernel */
    int exec_grid = 0, exec8_grid = 0;
    uint32_t* fallback_list = nullptr;                                  /* n entries, allocated with frame_first */
    int last_grid = 0;
    int last_hip_error = 0, last_hip_line = 0;
    /* staging for cz_decode_batch_host */
    void* d_stage = nullptr; size_t d_stage_bytes = 0;
    void* h_pin = nullptr; size_t h_pin_bytes = 0;                      /* pinned host staging of cz_decode_batch_multi's share */
    unsigned long long* d_prof = nullptr;   /* CZ_PROFILE builds: per-phase cycle sums */
    /* optional FSE-chain pre-pass */
    uint64_t* chain_arena = nullptr; uint64_t chain_capacity = 0;   /* 8-byte units */
    unsigned long long* chain_top = nullptr; uint32_t* chain_counter = nullptr;
    uint64_t* frame_first = nullptr; size_t frame_first_cap = 0;
    cz_blk_desc* blk_desc = nullptr; uint32_t blk_capacity = 0; uint32_t* scan_ctl = nullptr;   /* block list of the pre-pass */
    uint32_t* frame_order = nullptr;                                    /* n entries, allocated with frame_first */
    uint32_t* scan_wave = nullptr;                                      /* 72 words per wave of cz_scan_kernel, allocated with frame_first */
    int chain_grid = 0; uint32_t chain_min_nseq = 0;
    /* optional: block-parallel huff0 and tile kernels next to / behind the chain kernel, on streams of their own */
    uint8_t* lit_arena = nullptr; uint64_t lit_capacity = 0; unsigned long long* lit_top = nullptr;
    uint64_t* lit_first = nullptr; size_t lit_first_cap = 0; uint32_t* lit_counter = nullptr;
    hipStream_t stream2 = nullptr, stream3 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr, ev_lit = nullptr; int huf_grid = 0, huf1_grid = 0, tile_grid = 0; bool timed_lit = false;
    cz_lit_seg* lit_segs = nullptr; cz_copy_seg* copy_segs = nullptr; uint32_t seg_capacity = 0;   /* lists of the literal / copy pre-pass (cz_scan_kernel) */
    uint32_t* frame_pre = nullptr;                                      /* n entries, allocated with lit_first */
    uint32_t verify_checksum = 0;
    /* A launch that repeats the one before it — same arguments, same context settings — is captured as a hipGraph and replayed
       (cz_launch): one submission instead of ~35 stream operations per batch. */
    int graph_mode = 0;                    /* cz_context_set_graph_replay: 0 never (default), 1 from the second identical launch on */
    uint64_t cfg_gen = 0;                  /* grows with every change of the context that a launch depends on */
    hipStream_t gstream = nullptr;         /* the capture's origin stream (the caller's may be the NULL stream, which cannot capture) */
    hipGraphExec_t g_exec = nullptr; cz_batch_args g_proto; size_t g_n = 0; uint64_t g_gen = 0;   /* the captured launch */
    cz_batch_args g_seen_proto; size_t g_seen_n = 0; uint64_t g_seen_gen = 0; bool g_seen = false;   /* the last launch that went the ordinary way */
    bool g_bad = false;                    /* capture failed once on this context: not tried again */
    bool capturing = false, g_replayed = false;
    bool g_timed_chain = false, g_timed_exec = false, g_timed_lit = false, g_timed_wx = false, g_timed_small = false; int g_grid = 0;
    hipEvent_t ev_lit_dep = nullptr, ev_small_dep = nullptr;   /* inside a capture the timed events are external record nodes; waits go through these */
};

#define CZ_HIP(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) { (ctx)->last_hip_error = (int)_e; (ctx)->last_hip_line = __LINE__; if (getenv("CZ_GRAPH_DEBUG")) fprintf(stderr, "CZ_HIP: error %d at line %d (capturing %d)\n", (int)_e, __LINE__, (int)(ctx)->capturing); return CZ_E_HIP; } } while (0)

CZ_EXPORT int cz_abi_version(void) { return CZ_ABI_VERSION; }
CZ_EXPORT void cz_context_destroy(cz_context* c);

CZ_EXPORT int cz_context_create(cz_context** out, int device, void* stream) {
    if (!out) return CZ_E_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return CZ_E_NO_DEVICE;
    cz_context* c = new (std::nothrow) cz_context();
    if (!c) return CZ_E_INVALID_ARG;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return CZ_E_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return CZ_E_NO_DEVICE; }
    if (!strstr(prop.gcnArchName, "gfx950")) { delete c; return CZ_E_NO_DEVICE; }   /* kernels are built for gfx950 only */
    c->num_cu = prop.multiProcessorCount;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cz_decode_frames_kernel, CZ_WG_THREADS, CZ_MAIN_DYN_LDS) != hipSuccess || occ <= 0) occ = 4;
    c->occupancy = occ; c->grid_max = c->num_cu * occ;
    if (stream) c->stream = (hipStream_t)stream;
    else { if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { c->stream = nullptr; delete c; return CZ_E_HIP; } c->own_stream = true; }
    /* the per-workgroup literal scratch (266 KB each) is allocated by the first launch, for the workgroups it uses */
    /* every counter the kernels of a launch share — work_counter (64 bytes), chain_top (64), lit_top (64), scan_ctl — in ONE block, cleared by
       one memset per launch (four separate ones were four tiny kernels and their boundaries in front of every step) */
    if (hipMalloc((void**)&c->work_counter, CZ_CTL_BLOCK_BYTES) != hipSuccess ||
        hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_mid) != hipSuccess || hipEventCreate(&c->ev_mid2) != hipSuccess ||
        hipEventCreate(&c->ev_stop) != hipSuccess) {
        cz_context_destroy(c); return CZ_E_HIP;
    }
#ifdef CZ_PROFILE
    if (hipMalloc((void**)&c->d_prof, 64 * 8) == hipSuccess) (void)hipMemset(c->d_prof, 0, 64 * 8);
#endif
    *out = c;
    return CZ_OK;
}

/* Diagnostic builds (-DCZ_PROFILE): copies out and clears the per-phase cycle sums; returns the
 * number of phases, 0 in the product build. */
CZ_EXPORT int cz_context_read_profile(cz_context* c, unsigned long long* out, int cap) {
    if (!c || !c->d_prof) return 0;
    unsigned long long tmp[64];
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    if (hipMemcpy(tmp, c->d_prof, sizeof tmp, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    (void)hipMemset(c->d_prof, 0, sizeof tmp);
    int n = CZ_P_COUNT < cap ? CZ_P_COUNT : cap;
    for (int i = 0; i < n; i++) out[i] = tmp[i];
    for (int i = 20; i < 58 && i < cap; i++) out[i] = tmp[i];        /* (20..30: unused since the literals pass went); cz_chain_kernel: see CZC_PROF_* */
    return n;
}

CZ_EXPORT void cz_context_destroy(cz_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->lit_scratch) (void)hipFree(c->lit_scratch);
    if (c->work_counter) (void)hipFree(c->work_counter);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->d_prof) (void)hipFree(c->d_prof);
    if (c->chain_arena) (void)hipFree(c->chain_arena);
    if (c->lit_arena) (void)hipFree(c->lit_arena);
    if (c->lit_first) (void)hipFree(c->lit_first);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_join3) (void)hipEventDestroy(c->ev_join3);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->lit_segs) (void)hipFree(c->lit_segs);
    if (c->copy_segs) (void)hipFree(c->copy_segs);
    if (c->frame_pre) (void)hipFree(c->frame_pre);
    if (c->ev_lit) (void)hipEventDestroy(c->ev_lit);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->frame_first) (void)hipFree(c->frame_first);
    if (c->blk_desc) (void)hipFree(c->blk_desc);
    if (c->frame_order) (void)hipFree(c->frame_order);
    if (c->scan_wave) (void)hipFree(c->scan_wave);
    if (c->fallback_list) (void)hipFree(c->fallback_list);
    if (c->wx_list) (void)hipFree(c->wx_list);
    if (c->ev_wx) (void)hipEventDestroy(c->ev_wx);
    if (c->ev_small) (void)hipEventDestroy(c->ev_small);
    if (c->ev_e1) (void)hipEventDestroy(c->ev_e1);
    if (c->ev_w1) (void)hipEventDestroy(c->ev_w1);
    if (c->ev_x4) (void)hipEventDestroy(c->ev_x4);
    if (c->g_exec) (void)hipGraphExecDestroy(c->g_exec);
    if (c->gstream) (void)hipStreamDestroy(c->gstream);
    if (c->ev_lit_dep) (void)hipEventDestroy(c->ev_lit_dep);
    if (c->ev_small_dep) (void)hipEventDestroy(c->ev_small_dep);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_mid) (void)hipEventDestroy(c->ev_mid);
    if (c->ev_mid2) (void)hipEventDestroy(c->ev_mid2);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

CZ_EXPORT int cz_context_synchronize(cz_context* c) {
    if (!c) return CZ_E_INVALID_ARG;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    return CZ_OK;
}
CZ_EXPORT int cz_context_last_hip_error(const cz_context* c) { return c ? c->last_hip_error : 0; }
CZ_EXPORT int cz_context_launch_info(const cz_context* c, int* workgroups, int* threads_per_wg, int* compute_units) {
    if (!c) return CZ_E_INVALID_ARG;
    if (workgroups) *workgroups = c->last_grid ? c->last_grid : c->grid_max;
    if (threads_per_wg) *threads_per_wg = CZ_WG_THREADS;
    if (compute_units) *compute_units = c->num_cu;
    return CZ_OK;
}
/* Waves cz_execute_frames_kernel / cz_execute_frames8_kernel are launched with (0 before the first cz_context_set_chain_arena). */
CZ_EXPORT int cz_context_execute_grid(const cz_context* c, int* waves, int* waves8) {
    if (!c) return CZ_E_INVALID_ARG;
    if (waves) *waves = c->exec_grid;
    if (waves8) *waves8 = c->exec8_grid;
    return CZ_OK;
}
CZ_EXPORT int cz_context_last_kernel_ms(cz_context* c, float* ms) {
    if (!c || !ms || !c->timed) return CZ_E_INVALID_ARG;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipEventSynchronize(c->ev_stop));
    CZ_HIP(c, hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return CZ_OK;
}
/* Part of that launch spent in cz_chain_kernel (0 when the pre-pass is off). */
CZ_EXPORT int cz_context_last_chain_ms(cz_context* c, float* ms) {
    if (!c || !ms) return CZ_E_INVALID_ARG;
    *ms = 0.0f;
    if (!c->timed || !c->timed_chain) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipEventSynchronize(c->ev_stop));
    CZ_HIP(c, hipEventElapsedTime(ms, c->ev_start, c->ev_mid));
    return CZ_OK;
}

/* Enables (bytes > 0) or disables (0) the FSE-chain pre-pass for batch decodes on this context and
 * sizes its record arena: 8 bytes per sequence + 1312 per block with sequences; frames that do not fit fall back
 * to in-kernel chains, so any size is safe. */
CZ_EXPORT int cz_context_set_chain_arena(cz_context* c, size_t bytes) {
    if (!c) return CZ_E_INVALID_ARG;
    c->cfg_gen++;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    if (c->chain_arena) { (void)hipFree(c->chain_arena); c->chain_arena = nullptr; c->chain_capacity = 0; }
    if (!bytes) return CZ_OK;
    if (bytes < 4096) bytes = 4096;                                     /* header indices 0..63 are reserved (sink of the chain step) */
    if (!c->chain_top) { c->chain_top = (unsigned long long*)((uint8_t*)c->work_counter + 64); c->chain_counter = (uint32_t*)((uint8_t*)c->chain_top + 16); }
    if (!c->exec_grid) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, czx::cz_execute_frames_kernel, CZ_WG_THREADS, CZ_EXEC_DYN_LDS) != hipSuccess || occ <= 0) occ = 4;
        c->exec_grid = c->num_cu * occ;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, czx8::cz_execute_frames8_kernel, CZ_WG_THREADS, CZ_EXEC_DYN_LDS) != hipSuccess || occ <= 0) occ = 8;
        c->exec8_grid = c->num_cu * occ;
    }
    CZ_HIP(c, hipMalloc((void**)&c->chain_arena, (bytes + 7) & ~(size_t)7));
    c->chain_capacity = bytes / 8;
    /* every listed block takes at least 4 + CZ_CHAIN_MAP_WORDS + 1 arena units: that bounds the block list */
    if (c->blk_desc) { (void)hipFree(c->blk_desc); c->blk_desc = nullptr; }
    c->blk_capacity = (uint32_t)(c->chain_capacity / (4 + CZ_CHAIN_MAP_WORDS + 1) + 4096);
    CZ_HIP(c, hipMalloc((void**)&c->blk_desc, (size_t)c->blk_capacity * sizeof(cz_blk_desc)));
    if (!c->scan_ctl) c->scan_ctl = (uint32_t*)((uint8_t*)c->work_counter + 192);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cz_chain_kernel, CZ_WG_THREADS, 0) != hipSuccess || occ <= 0) occ = 2;
    c->chain_grid = c->num_cu * occ;
    return CZ_OK;
}

/* Enables (bytes > 0) or disables (0) the literal / copy half of the pre-pass: with the chain arena set, cz_scan_kernel also lists
 * every Huffman-coded literals section and every Raw / RLE run whose place in the output follows from the headers;
 * cz_huf_kernel (unit of work: one section) and cz_tile_kernel do them NEXT TO cz_chain_kernel, on streams of their own.
 * Literals of blocks that have sequences go to nodes of the literal arena (bytes = its capacity: decoded literal bytes + 16 per
 * block, never more than the decoded size of the batch); literals of blocks without sequences ahead of a frame's first block
 * with sequences, like the Raw / RLE runs, go straight into the output.  Frames that do not fit, or are irregular in any way,
 * are decoded entirely by cz_decode_frames_kernel. */
CZ_EXPORT int cz_context_set_literal_arena(cz_context* c, size_t bytes) {
    if (!c) return CZ_E_INVALID_ARG;
    c->cfg_gen++;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    if (c->lit_arena) { (void)hipFree(c->lit_arena); c->lit_arena = nullptr; c->lit_capacity = 0; }
    if (c->lit_segs) { (void)hipFree(c->lit_segs); c->lit_segs = nullptr; }
    if (c->copy_segs) { (void)hipFree(c->copy_segs); c->copy_segs = nullptr; }
    c->seg_capacity = 0;
    if (!bytes) return CZ_OK;
    if (bytes < 4096) bytes = 4096;
    if (!c->lit_top) { c->lit_top = (unsigned long long*)((uint8_t*)c->work_counter + 128); c->lit_counter = (uint32_t*)((uint8_t*)c->lit_top + 16); }
    if (!c->stream2) {
        CZ_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        CZ_HIP(c, hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
        CZ_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        CZ_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        CZ_HIP(c, hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming));
        CZ_HIP(c, hipEventCreate(&c->ev_lit));
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cz_huf_kernel, CZH_THREADS, 0) != hipSuccess || occ <= 0) occ = 4;
        c->huf_grid = c->num_cu * occ;
        c->tile_grid = c->num_cu * 8;
        c->huf1_grid = c->num_cu * 4;                                   /* what fits on a CU next to four chain waves */
    }
    CZ_HIP(c, hipMalloc((void**)&c->lit_arena, (bytes + 15) & ~(size_t)15));
    c->lit_capacity = bytes;
    /* one list entry per Huffman section / per run: bounded by the arena for sections that take a node (>= 32 bytes each); the
       others are bounded here by one entry per 256 bytes of arena — a frame whose entries do not fit is simply not pre-passed */
    size_t cap = bytes / 256 + 65536; if (cap > (1u << 24)) cap = 1u << 24;
    CZ_HIP(c, hipMalloc((void**)&c->lit_segs, cap * sizeof(cz_lit_seg)));
    CZ_HIP(c, hipMalloc((void**)&c->copy_segs, cap * sizeof(cz_copy_seg)));
    c->seg_capacity = (uint32_t)cap;
    return CZ_OK;
}

/* Frames the pre-pass finished run on cz_execute_frames_kernel (default, 1) or, like every other frame, on cz_decode_frames_kernel (0). */
CZ_EXPORT int cz_context_set_exec_kernel(cz_context* c, int on) { if (!c) return CZ_E_INVALID_ARG; c->cfg_gen++; c->exec_kernel = on != 0; c->exec_variant_force = on == 4 || on == 8 ? (uint32_t)on : 0u; return CZ_OK; }

CZ_EXPORT int cz_context_set_debug_flags(cz_context* c, uint32_t flags) { if (!c) return CZ_E_INVALID_ARG; c->cfg_gen++; c->debug_flags = flags; return CZ_OK; }
CZ_EXPORT int cz_context_debug_read_chain_arena(cz_context* c, void* dst, size_t bytes, uint64_t* units_in_use) {
    if (!c || (bytes && !dst)) return CZ_E_INVALID_ARG;
    if (units_in_use) *units_in_use = 0;
    if (!c->chain_arena) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    if (bytes > c->chain_capacity * 8) bytes = c->chain_capacity * 8;
    if (bytes) CZ_HIP(c, hipMemcpy(dst, c->chain_arena, bytes, hipMemcpyDeviceToHost));
    if (units_in_use) { unsigned long long top = 0; CZ_HIP(c, hipMemcpy(&top, c->chain_top, 8, hipMemcpyDeviceToHost)); *units_in_use = 64ull + top; }
    return CZ_OK;
}
CZ_EXPORT int cz_context_set_verify_checksum(cz_context* c, int on) { if (!c) return CZ_E_INVALID_ARG; c->cfg_gen++; c->verify_checksum = on ? 1u : 0u; return CZ_OK; }

/* Frames whose first sequences section has fewer sequences than this are not pre-passed (default 0: every frame that has
 * sequences is; with the block-parallel pre-pass that measured fastest on the corpus-like mix too). */
CZ_EXPORT int cz_context_set_chain_min_sequences(cz_context* c, uint32_t n) { if (!c) return CZ_E_INVALID_ARG; c->cfg_gen++; c->chain_min_nseq = n; return CZ_OK; }

/* Diagnostics of the last batch launch (synchronises): how many of its n frames got chain records from the pre-pass,
 * and how many had their literals decoded by the huff0 kernels. */
CZ_EXPORT int cz_context_last_prepass_counts(cz_context* c, size_t n, size_t* with_chain, size_t* with_literals) {
    if (!c) return CZ_E_INVALID_ARG;
    if (with_chain) *with_chain = 0;
    if (with_literals) *with_literals = 0;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<uint64_t> h(n);
    if (with_chain && c->chain_arena && c->frame_first && n <= c->frame_first_cap) {
        CZ_HIP(c, hipMemcpy(h.data(), c->frame_first, n * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) *with_chain += h[i] != 0;
    }
    if (with_literals && c->lit_arena && c->lit_first && n <= c->lit_first_cap) {
        CZ_HIP(c, hipMemcpy(h.data(), c->lit_first, n * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) *with_literals += h[i] != 0;
    }
    return CZ_OK;
}

/* How long the last launch went on with cz_huf_kernel / cz_huf1_kernel / cz_tile_kernel after cz_chain_kernel was done (0: no literal arena). */
CZ_EXPORT int cz_context_last_literals_tail_ms(cz_context* c, float* ms) {
    if (!c || !ms) return CZ_E_INVALID_ARG;
    *ms = 0.0f;
    if (!c->timed || !c->timed_lit || !c->timed_chain) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipEventSynchronize(c->ev_stop));
    CZ_HIP(c, hipEventElapsedTime(ms, c->ev_mid, c->ev_lit));
    return CZ_OK;
}

/* The chain pre-pass as two launches — large blocks / all others — with the early execute launches behind the second (1), or one
   launch and the execute stage behind all of it (0, the default: on the corpus-like mix the small blocks' launch takes as long as
   the large blocks' — table parse and build per block, not chain steps — so nothing is ready early; profiles/r5/NOTES.md). */
CZ_EXPORT int cz_context_set_early_execute(cz_context* c, int on) { if (!c) return CZ_E_INVALID_ARG; c->cfg_gen++; c->early_execute = on != 0; return CZ_OK; }
/* When the small blocks' chains and every literal of the last launch were done, in ms from its start (0: not a split launch). */
CZ_EXPORT int cz_context_last_small_ms(cz_context* c, float* ms) {
    if (!c || !ms) return CZ_E_INVALID_ARG;
    *ms = 0.0f;
    if (!c->timed || !c->timed_small) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipEventSynchronize(c->ev_stop));
    CZ_HIP(c, hipEventElapsedTime(ms, c->ev_start, c->ev_small));
    return CZ_OK;
}
/* Far-offset batches run cz_wexec_kernel side by side with cz_execute_frames_kernel (default, 1), or cz_execute_frames_kernel alone (0). */
CZ_EXPORT int cz_context_set_wexec_kernel(cz_context* c, int on) { if (!c) return CZ_E_INVALID_ARG; c->cfg_gen++; c->wexec_kernel = on != 0; return CZ_OK; }
/* A/B knobs of the side-by-side execute stage: CUs cz_wexec_kernel runs on (0: half), frames per CU of it that cz_execute_frames_kernel
   leaves to it at the end of a batch (0: default), force = 1: side by side whatever the batch's offsets look like. */
CZ_EXPORT int cz_context_set_wexec_tuning(cz_context* c, int cus, int leave_per_cu, int force) {
    if (!c || cus < 0 || leave_per_cu < 0) return CZ_E_INVALID_ARG;
    c->cfg_gen++;
    c->wexec_cus = cus; if (leave_per_cu) c->wexec_leave_per_cu = leave_per_cu; c->wexec_force = force == 2 ? 2u : (force ? 1u : 0u);   /* 1: side by side whatever the offsets; 2: never the large frames of a near-offset batch alone (A/B) */
    return CZ_OK;
}
/* A batch launch that repeats the one before it (same pointers, sizes and context settings; the bytes may differ) is captured as a
   hipGraph and replayed from then on: 1; default 0 = every launch is enqueued operation by operation. */
CZ_EXPORT int cz_context_set_graph_replay(cz_context* c, int on) {
    if (!c) return CZ_E_INVALID_ARG;
    c->graph_mode = on != 0; c->g_seen = false; c->g_bad = false;
    if (!on && c->g_exec) { CZ_HIP(c, hipSetDevice(c->device)); CZ_HIP(c, hipStreamSynchronize(c->stream)); (void)hipGraphExecDestroy(c->g_exec); c->g_exec = nullptr; }
    return CZ_OK;
}
/* 1 when the most recent batch launch was the replay of a captured graph. */
CZ_EXPORT int cz_context_last_launch_was_replay(const cz_context* c) { return c && c->g_replayed ? 1 : 0; }
/* Diagnostics of the most recent batch launch (synchronises): what cz_chain_kernel summed from the blocks' code tables, in sequences
   x 4 — with near offset codes (2..13), with far ones (14 and up), with a literal run above 8 or a match above 16 bytes: what
   cz_wx_side_by_side and cz_exec_variant decide from. */
CZ_EXPORT int cz_context_last_sequence_stats(cz_context* c, uint64_t* near_offsets, uint64_t* far_offsets, uint64_t* long_runs) {
    if (!c) return CZ_E_INVALID_ARG;
    uint64_t h[3] = {0, 0, 0};
    if (c->chain_top) {
        CZ_HIP(c, hipSetDevice(c->device));
        CZ_HIP(c, hipStreamSynchronize(c->stream));
        CZ_HIP(c, hipMemcpy(h, c->chain_top + 5, sizeof h, hipMemcpyDeviceToHost));
    }
    if (near_offsets) *near_offsets = h[0];
    if (far_offsets) *far_offsets = h[1];
    if (long_runs) *long_runs = h[2];
    return CZ_OK;
}
/* Diagnostics of the most recent batch launch (synchronises): frames listed for cz_wexec_kernel, frames it finished, frames it gave up. */
CZ_EXPORT int cz_context_last_wexec_counts(cz_context* c, size_t* listed, size_t* finished, size_t* given_up) {
    if (!c) return CZ_E_INVALID_ARG;
    if (listed) *listed = 0;
    if (finished) *finished = 0;
    if (given_up) *given_up = 0;
    if (!c->scan_ctl) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t h[4] = {0, 0, 0, 0};
    CZ_HIP(c, hipMemcpy(h, c->scan_ctl + 206, sizeof h, hipMemcpyDeviceToHost));
    if (getenv("CZ_SIDE_COUNTS")) { uint32_t g[4] = {0, 0, 0, 0}; (void)hipMemcpy(g, c->scan_ctl + 213, sizeof g, hipMemcpyDeviceToHost); fprintf(stderr, "side counts: wexec workgroups counted in %u, (early %u), execute waves that left %u, that waited and stayed %u\n", g[0], g[1], g[2], g[3]); }
    if (listed) *listed = h[0];
    if (given_up) *given_up = h[1];
    if (finished) *finished = h[3];
    return CZ_OK;
}
/* Diagnostics of the most recent batch launch (synchronises): entries on the fall-back list, i.e. frames the pre-pass and execute
   kernels handed to cz_decode_frames_kernel (0 without the execute stage, where that kernel takes all n frames anyway). */
CZ_EXPORT int cz_context_last_fallback_count(cz_context* c, size_t* listed) {
    if (!c || !listed) return CZ_E_INVALID_ARG;
    *listed = 0;
    if (!c->chain_top || !c->fallback_list) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t h = 0;
    CZ_HIP(c, hipMemcpy(&h, (uint8_t*)c->chain_top + 28, 4, hipMemcpyDeviceToHost));
    *listed = h;
    return CZ_OK;
}
/* Part of the last launch spent in cz_wexec_kernel (0 when it did not run). */
CZ_EXPORT int cz_context_last_wexec_ms(cz_context* c, float* ms) {
    if (!c || !ms) return CZ_E_INVALID_ARG;
    *ms = 0.0f;
    if (!c->timed || !c->timed_wx) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipEventSynchronize(c->ev_stop));
    CZ_HIP(c, hipEventElapsedTime(ms, c->timed_lit ? c->ev_lit : c->ev_mid, c->ev_wx));
    return CZ_OK;
}

/* Part of the last launch spent in cz_execute_frames_kernel (0 when it did not run). */
CZ_EXPORT int cz_context_last_exec_ms(cz_context* c, float* ms) {
    if (!c || !ms) return CZ_E_INVALID_ARG;
    *ms = 0.0f;
    if (!c->timed || !c->timed_exec) return CZ_OK;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipEventSynchronize(c->ev_stop));
    CZ_HIP(c, hipEventElapsedTime(ms, c->timed_lit ? c->ev_lit : c->ev_mid, c->ev_mid2));
    return CZ_OK;
}

/* ------------------------------------------------------------------ launch */
#define CZ_E_NOGRAPH (-32000)            /* internal: this launch needs something a stream capture cannot hold (an allocation) */
/* Enqueues one batch launch on `s0` and the context's other streams.  With c->capturing, s0 is the origin stream of a capture:
   nothing may allocate or synchronise (CZ_E_NOGRAPH), and the events the cz_context_last_*_ms calls read are recorded as external
   event nodes, so that a replay of the graph stamps them again. */
static int cz_enqueue(cz_context* c, const cz_batch_args& proto, size_t n, hipStream_t s0) {
    const bool cap = c->capturing;
    /* a timed event: inside a capture an event-record NODE behind the stream's last captured nodes, which then takes their place */
    auto rec_t = [&](hipEvent_t ev, hipStream_t st) -> hipError_t {
        if (!cap) return hipEventRecord(ev, st);
        hipStreamCaptureStatus status = hipStreamCaptureStatusNone; unsigned long long id = 0; hipGraph_t g = nullptr; const hipGraphNode_t* deps = nullptr; size_t ndeps = 0;
        hipError_t e = hipStreamGetCaptureInfo_v2(st, &status, &id, &g, &deps, &ndeps);
        if (e != hipSuccess) return e;
        if (status != hipStreamCaptureStatusActive || !g) return hipErrorStreamCaptureInvalidated;
        hipGraphNode_t node = nullptr;
        e = hipGraphAddEventRecordNode(&node, g, deps, ndeps, ev);
        if (e != hipSuccess) return e;
        return hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
    };
    cz_batch_args a = proto;
    a.n = (uint32_t)n; a.work_counter = c->work_counter; a.lit_scratch = c->lit_scratch; a.lit_scratch_stride = CZ_WG_SCRATCH_BYTES;
    a.prof = c->d_prof; a.verify_checksum = a.tasks ? 0 : c->verify_checksum; a.debug_flags = c->debug_flags;
    if (!a.tasks && c->batch_dict) { a.dict_state = c->batch_dict->d_state; a.dict = c->batch_dict->d_raw + c->batch_dict->content_off; a.dict_len = c->batch_dict->len - c->batch_dict->content_off; }
    int grid = (int)(n < (size_t)c->grid_max ? n : (size_t)c->grid_max);
#ifdef CZ_EXPERIMENT
    if (const char* e = getenv("CZ_GRID_PER_CU")) { const int g = atoi(e) * c->num_cu; if (g > 0 && g < grid) grid = g; }
#endif
    if (c->lit_slots < grid) {                                          /* one literal scratch region per resident workgroup */
        if (cap) return CZ_E_NOGRAPH;
        c->cfg_gen++;
        if (c->lit_scratch) { CZ_HIP(c, hipStreamSynchronize(s0)); (void)hipFree(c->lit_scratch); c->lit_scratch = nullptr; c->lit_slots = 0; }
        const int slots = grid <= 16 ? 16 : c->grid_max;
        CZ_HIP(c, hipMalloc((void**)&c->lit_scratch, (size_t)slots * CZ_WG_SCRATCH_BYTES)); c->lit_slots = slots;
        a.lit_scratch = c->lit_scratch;
    }
    CZ_HIP(c, hipMemsetAsync(c->work_counter, 0, c->chain_arena && !proto.tasks ? CZ_CTL_BLOCK_BYTES : 64, s0));   /* the whole control block */
    if (!cap) CZ_HIP(c, hipEventRecord(c->ev_start, s0));   /* (a replayed graph: cz_launch records it in front of the graph) */
    a.chain_arena = nullptr; a.chain_capacity = 0; a.chain_top = nullptr; a.frame_first = nullptr; a.chain_counter = nullptr;
    if (c->chain_arena && !a.tasks) {
        /* the pre-pass: block list (cz_scan_kernel), then the FSE chains of all blocks (cz_chain_kernel) -> records in the arena */
        if (c->frame_first_cap < n) {
            if (cap) return CZ_E_NOGRAPH;
            c->cfg_gen++;
            if (c->frame_first) { CZ_HIP(c, hipStreamSynchronize(s0)); (void)hipFree(c->frame_first); c->frame_first = nullptr; c->frame_first_cap = 0; }
            CZ_HIP(c, hipMalloc((void**)&c->frame_first, n * 8));        /* (frame_first_cap is set once every list below is there: a failure in between leaves the context asking again) */
            if (c->frame_order) (void)hipFree(c->frame_order);
            c->frame_order = nullptr;
            CZ_HIP(c, hipMalloc((void**)&c->frame_order, n * 4));
            if (c->scan_wave) (void)hipFree(c->scan_wave);
            c->scan_wave = nullptr;
            CZ_HIP(c, hipMalloc((void**)&c->scan_wave, ((n + CZ_WG_THREADS - 1) / CZ_WG_THREADS) * 72 * 4));
            if (c->fallback_list) (void)hipFree(c->fallback_list);
            c->fallback_list = nullptr;
            CZ_HIP(c, hipMalloc((void**)&c->fallback_list, n * 4));
            if (c->wx_list) (void)hipFree(c->wx_list);
            c->wx_list = nullptr;
            CZ_HIP(c, hipMalloc((void**)&c->wx_list, n * 4));
            c->frame_first_cap = n;
        }
        a.chain_arena = c->chain_arena; a.chain_capacity = c->chain_capacity; a.chain_top = c->chain_top;
        a.frame_first = c->frame_first; a.chain_counter = c->chain_counter; a.chain_min_nseq = c->chain_min_nseq;
        const bool lit_pass = c->lit_arena != nullptr;
        if (lit_pass) {
            if (c->lit_first_cap < n) {
                if (cap) return CZ_E_NOGRAPH;
                c->cfg_gen++;
                if (c->lit_first) { CZ_HIP(c, hipStreamSynchronize(s0)); (void)hipFree(c->lit_first); c->lit_first = nullptr; c->lit_first_cap = 0; }
                CZ_HIP(c, hipMalloc((void**)&c->lit_first, n * 8)); c->lit_first_cap = n;
                if (c->frame_pre) (void)hipFree(c->frame_pre);
                c->frame_pre = nullptr;
                CZ_HIP(c, hipMalloc((void**)&c->frame_pre, n * 4));
            }
            a.lit_arena = c->lit_arena; a.lit_capacity = c->lit_capacity; a.lit_top = c->lit_top; a.lit_first = c->lit_first;
            a.lit_segs = c->lit_segs; a.lit_seg_capacity = c->seg_capacity; a.copy_segs = c->copy_segs; a.copy_seg_capacity = c->seg_capacity; a.frame_pre = c->frame_pre;
        }
        /* which kernels the execute stage has is settled BEFORE the scan, so that every kernel of the batch sees the same wx_list */
        const bool use_exec = c->exec_kernel && lit_pass && !c->batch_dict;
        bool use_wx = use_exec && c->wexec_kernel;
        if (use_wx && !c->wexec_ready) {
            if (cap) return CZ_E_NOGRAPH;
            c->cfg_gen++;
            /* (a workgroup of cz_wexec_kernel asks for more LDS than the default limit: where the runtime will not grant it,
               cz_execute_frames_kernel does the whole batch, as with cz_context_set_wexec_kernel(ctx, 0)) */
            if (hipFuncSetAttribute((const void*)cz_wexec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WX_LDS_BYTES) != hipSuccess ||
                hipEventCreate(&c->ev_wx) != hipSuccess) { (void)hipGetLastError(); c->wexec_kernel = false; use_wx = false; }
            else c->wexec_ready = true;
        }
        if (use_wx) { a.wx_list = c->wx_list; a.wx_counter = (uint32_t*)((uint8_t*)c->chain_top + 20); }
        /* pass A0: the block list (cz_scan_kernel, one lane per frame, two passes: count, place) */
        a.blk_desc = c->blk_desc; a.blk_capacity = c->blk_capacity; a.scan_ctl = c->scan_ctl; a.frame_order = c->frame_order; a.scan_wave = c->scan_wave;
        if (use_exec) { a.exec_counter = (uint32_t*)((uint8_t*)c->chain_top + 24); a.fallback_count = (uint32_t*)((uint8_t*)c->chain_top + 28); a.fallback_list = c->fallback_list; }
        const int sgrid = (int)((n + CZ_WG_THREADS - 1) / CZ_WG_THREADS);
        a.scan_pass = 0; hipLaunchKernelGGL(cz_scan_kernel, dim3(sgrid), dim3(CZ_WG_THREADS), 0, s0, a);
        a.scan_pass = 1; hipLaunchKernelGGL(cz_scan_kernel, dim3(sgrid), dim3(CZ_WG_THREADS), 0, s0, a);
        CZ_HIP(c, hipGetLastError());
        /* With the execute stage on, the chain pre-pass is TWO launches (czstd_types.h, CZ_BIG_BLOCK_SEQS): the large blocks on the
           context's stream — the batch lasts as long as its longest chain —, all others on a stream of their own, followed there by
           cz_huf_kernel and the joins of the literal kernels.  When THAT stream is done (ev_small), every frame without a large block
           is ready, and two early launches start beside the large blocks' chains: cz_execute_frames_kernel for those frames, and
           cz_wexec_kernel for the batch's large frames, block by block behind their chains' flags.  The launches behind the large
           chains are the ones there always were; frames are claimed, so whoever gets to a frame first does it. */
        const bool split = use_exec && c->early_execute;
        /* FOUR streams in all (the runtime multiplexes more than that onto four hardware queues, and two streams on one queue run one
           after the other): the context's — large chains, then cz_wexec_kernel's later launch; stream2 — cz_huf1_kernel, then
           cz_wexec_kernel's early launch; stream3 — cz_tile_kernel, then cz_execute_frames_kernel's early launch; stream4 — small
           chains, cz_huf_kernel, then cz_execute_frames_kernel's later launch. */
        if (split && !c->stream4) {
            if (cap) return CZ_E_NOGRAPH;
            c->cfg_gen++;
            if (hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev_small) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_e1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_w1, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_x4, hipEventDisableTiming) != hipSuccess) {
                c->last_hip_error = (int)hipGetLastError(); return CZ_E_HIP;
            }
        }
        /* the literal and copy kernels may start when the chain kernel does (not before: they would take the LDS the chain
           kernel's workgroups need and hold them up) */
#ifdef CZ_EXPERIMENT
        /* diagnostic (CZ_EXP_OVERLAP = workgroups per CU): what cz_execute_frames_kernel and cz_chain_kernel cost each other when they
           run side by side.  The execute kernel works on the records the PREVIOUS launch left in the arena (same batch, same
           bytes), after all literals; its output is the same bytes again.  Read the kernel trace, not the event times. */
        const char* ovl = use_exec ? getenv("CZ_EXP_OVERLAP") : nullptr;
        if (ovl) {
            hipLaunchKernelGGL(cz_huf_kernel, dim3(c->huf_grid), dim3(CZH_THREADS), 0, s0, a);
            CZ_HIP(c, hipEventRecord(c->ev_fork, s0));
            CZ_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            cz_batch_args a2 = a; a2.exec_counter = c->work_counter;
            hipLaunchKernelGGL(czx::cz_execute_frames_kernel, dim3(atoi(ovl) * c->num_cu), dim3(CZ_WG_THREADS), CZ_EXEC_DYN_LDS, c->stream2, a2);
            CZ_HIP(c, hipEventRecord(c->ev_join, c->stream2));
            a.chain_grid = (uint32_t)c->chain_grid;
            hipLaunchKernelGGL(cz_chain_kernel, dim3(c->chain_grid), dim3(CZ_WG_THREADS), 0, s0, a);
            CZ_HIP(c, hipStreamWaitEvent(s0, c->ev_join, 0));
            CZ_HIP(c, hipMemsetAsync(c->work_counter, 0, 4, s0));
            CZ_HIP(c, hipEventRecord(c->ev_fork, s0));
        }
#endif
        if (lit_pass) CZ_HIP(c, hipEventRecord(c->ev_fork, s0));
        const int cgrid = c->chain_grid;                                /* the waves take blocks off the list until it is empty */
        a.chain_grid = (uint32_t)(split ? 2 * cgrid : cgrid);           /* (cz_huf1_kernel stops when this many chain waves have counted themselves out) */
        a.chain_part = split ? 1u : 0u;
        hipLaunchKernelGGL(cz_chain_kernel, dim3(cgrid), dim3(CZ_WG_THREADS), 0, s0, a);
        CZ_HIP(c, hipGetLastError());
        CZ_HIP(c, rec_t(c->ev_mid, s0));
        hipStream_t sl = s0;                                     /* the stream the rest of the pre-pass is enqueued on */
        if (split) {
            sl = c->stream4;
            CZ_HIP(c, hipStreamWaitEvent(sl, c->ev_fork, 0));
            a.chain_part = 2u;
            hipLaunchKernelGGL(cz_chain_kernel, dim3(cgrid), dim3(CZ_WG_THREADS), 0, sl, a);
            CZ_HIP(c, hipGetLastError());
            a.chain_part = 0u;
        }
        if (lit_pass) {
            /* next to the chain kernel, on streams of their own: cz_huf1_kernel (one wave per literals section: what fits beside the
               chain kernel's LDS) and cz_tile_kernel; behind the chain kernel, with the whole chip: cz_huf_kernel for what is left */
            /* (cz_tile_kernel is submitted first: behind cz_huf1_kernel its 256-thread workgroups waited for register space beside that
               kernel's waves — up to 1.3 ms for 0.1 ms of copies on the corpus-like mix) */
            CZ_HIP(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
            hipLaunchKernelGGL(cz_tile_kernel, dim3(c->tile_grid), dim3(256), 0, c->stream3, a);
            CZ_HIP(c, hipGetLastError());
            CZ_HIP(c, hipEventRecord(c->ev_join3, c->stream3));
            CZ_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
            int h1grid = c->huf1_grid;
#ifdef CZ_EXPERIMENT
            if (const char* e = getenv("CZ_HUF1_PER_CU")) { const int g = atoi(e) * c->num_cu; if (g > 0) h1grid = g; }
#endif
            if (!(c->debug_flags & CZ_DEBUG_NO_HUF1)) hipLaunchKernelGGL(cz_huf1_kernel, dim3(h1grid), dim3(CZ_WG_THREADS), 0, c->stream2, a);
            CZ_HIP(c, hipGetLastError());
            CZ_HIP(c, hipEventRecord(c->ev_join, c->stream2));
            hipLaunchKernelGGL(cz_huf_kernel, dim3(c->huf_grid), dim3(CZH_THREADS), 0, sl, a);
            CZ_HIP(c, hipGetLastError());
            CZ_HIP(c, hipStreamWaitEvent(sl, c->ev_join, 0));
            CZ_HIP(c, hipStreamWaitEvent(sl, c->ev_join3, 0));
            if (split) {
                CZ_HIP(c, rec_t(c->ev_small, sl));                      /* the small blocks' chains and every literal: done */
                if (cap) CZ_HIP(c, hipEventRecord(c->ev_small_dep, sl));
                CZ_HIP(c, hipStreamWaitEvent(s0, cap ? c->ev_small_dep : c->ev_small, 0));
            }
            CZ_HIP(c, rec_t(c->ev_lit, s0));                     /* all of the pre-pass done */
            if (cap) CZ_HIP(c, hipEventRecord(c->ev_lit_dep, s0));
            c->timed_lit = true;
        } else c->timed_lit = false;
        c->timed_chain = true;
        c->timed_exec = false; c->timed_wx = false; c->timed_small = split;
        if (use_exec) {
            int egrid = (int)(n < (size_t)c->exec_grid ? n : (size_t)c->exec_grid);
#ifdef CZ_EXPERIMENT
            if (const char* e = getenv("CZ_EXEC_PER_CU")) { const int g = atoi(e) * c->num_cu; if (g > 0 && g < egrid) egrid = g; }
#endif
            const int egrid8 = (int)(n < (size_t)c->exec8_grid ? n : (size_t)c->exec8_grid);
            a.wx_leave = 0; a.exec_variant_force = c->exec_variant_force; a.wx_force = c->wexec_force;
            if (split) {
                /* the early launches, behind ev_small, beside the large blocks' chains; work counters of their own */
                cz_batch_args e = a;
                e.early = 1u;
                e.exec_counter = (uint32_t*)((uint8_t*)c->chain_top + 12);
                if (use_wx) {                                           /* (first: its workgroups need whole CUs) */
                    e.wx_counter = (uint32_t*)((uint8_t*)c->chain_top + 8); e.wx_cus = 0;
                    CZ_HIP(c, hipStreamWaitEvent(c->stream2, cap ? c->ev_small_dep : c->ev_small, 0));
                    hipLaunchKernelGGL(cz_wexec_kernel, dim3(c->num_cu), dim3(WX_THREADS), WX_LDS_BYTES, c->stream2, e);
                    CZ_HIP(c, hipGetLastError());
                    CZ_HIP(c, hipEventRecord(c->ev_w1, c->stream2));
                }
                CZ_HIP(c, hipStreamWaitEvent(c->stream3, cap ? c->ev_small_dep : c->ev_small, 0));
                hipLaunchKernelGGL(czx::cz_execute_frames_kernel, dim3(egrid), dim3(CZ_WG_THREADS), CZ_EXEC_DYN_LDS, c->stream3, e);
                CZ_HIP(c, hipGetLastError());
                CZ_HIP(c, hipEventRecord(c->ev_e1, c->stream3));
                a.early = 2u;
            }
            hipStream_t sx = split ? c->stream4 : c->stream2;           /* where cz_execute_frames_kernel's later launch goes when cz_wexec_kernel runs beside it */
            if (use_wx) {
                /* Two kernels execute the sequences side by side and share the frames (each claims a frame before it starts on it):
                   cz_wexec_kernel — a workgroup of 16 waves per frame, the block in hand in an LDS window: bound by instruction
                   issue — on `wexec_cus` CUs (a workgroup of it fills a CU), and cz_execute_frames_kernel — a wave per frame, match
                   sources from HBM: bound by the rate of random reads, which does not need every CU — on the others.  The first is
                   launched on this stream, the second on a stream of its own behind an event: so the first is dispatched first and
                   gets its CUs.  (If it does not, it finds every frame claimed when it starts: nothing is lost but the overlap.) */
                /* (args.wx_cus workgroups of cz_wexec_kernel stay; while they are not all in place, cz_execute_frames_kernel's waves keep off the
                   even CUs — cz_cu_side —, so the split does not depend on which kernel the dispatcher places first) */
                const int wcus = c->wexec_cus > 0 ? (c->wexec_cus > c->num_cu ? c->num_cu : c->wexec_cus) : c->num_cu / 2;
                const int wgrid = wcus + CZ_WX_SPARE_WGS;
                a.wx_cus = (uint32_t)wcus;
                a.wx_leave = a.wx_cus * (uint32_t)c->wexec_leave_per_cu;
                const bool exec_first = (c->debug_flags & CZ_DEBUG_EXEC_FIRST) != 0;   /* test knob: the other submission order */
                if (!exec_first) {
                    hipLaunchKernelGGL(cz_wexec_kernel, dim3(wgrid), dim3(WX_THREADS), WX_LDS_BYTES, s0, a);
                    CZ_HIP(c, hipGetLastError());
                    CZ_HIP(c, rec_t(c->ev_wx, s0));
                }
                CZ_HIP(c, hipStreamWaitEvent(sx, cap ? c->ev_lit_dep : c->ev_lit, 0));
                hipLaunchKernelGGL(czx::cz_execute_frames_kernel, dim3(egrid), dim3(CZ_WG_THREADS), CZ_EXEC_DYN_LDS, sx, a);
                hipLaunchKernelGGL(czx8::cz_execute_frames8_kernel, dim3(egrid8), dim3(CZ_WG_THREADS), CZ_EXEC_DYN_LDS, sx, a);   /* (only one of the two builds does anything) */
                CZ_HIP(c, hipGetLastError());
                CZ_HIP(c, hipEventRecord(split ? c->ev_x4 : c->ev_join, sx));
                if (exec_first) {
                    CZ_HIP(c, hipEventRecord(c->ev_fork, sx));          /* (free by now: an event behind the other kernel's SUBMISSION is the best a host can do to put this one second) */
                    CZ_HIP(c, hipStreamWaitEvent(s0, cap ? c->ev_lit_dep : c->ev_lit, 0));
                    hipLaunchKernelGGL(cz_wexec_kernel, dim3(wgrid), dim3(WX_THREADS), WX_LDS_BYTES, s0, a);
                    CZ_HIP(c, hipGetLastError());
                    CZ_HIP(c, rec_t(c->ev_wx, s0));
                }
                c->timed_wx = true;
                CZ_HIP(c, hipStreamWaitEvent(s0, split ? c->ev_x4 : c->ev_join, 0));
            } else {
                /* the frames the pre-pass finished: cz_execute_frames_kernel (no decoders: 3 KB of LDS per wave and registers of its
                   own); it lists every frame it cannot do for cz_decode_frames_kernel */
                hipLaunchKernelGGL(czx::cz_execute_frames_kernel, dim3(egrid), dim3(CZ_WG_THREADS), CZ_EXEC_DYN_LDS, s0, a);
                hipLaunchKernelGGL(czx8::cz_execute_frames8_kernel, dim3(egrid8), dim3(CZ_WG_THREADS), CZ_EXEC_DYN_LDS, s0, a);   /* (only one of the two builds does anything) */
                CZ_HIP(c, hipGetLastError());
            }
            if (split) {                                                /* the early launches may outlast these */
                CZ_HIP(c, hipStreamWaitEvent(s0, c->ev_e1, 0));
                if (use_wx) CZ_HIP(c, hipStreamWaitEvent(s0, c->ev_w1, 0));
            }
            CZ_HIP(c, rec_t(c->ev_mid2, s0));
            c->timed_exec = true;
        }
    } else { c->timed_chain = false; c->timed_exec = false; c->timed_lit = false; c->timed_wx = false; c->timed_small = false; }
    /* (A launch of the record-consuming frames without the FSE tables in LDS was measured: the
       kernel is VGPR-limited to 16 waves per CU either way, so one launch serves all frames.) */
    hipLaunchKernelGGL(cz_decode_frames_kernel, dim3(grid), dim3(CZ_WG_THREADS), CZ_MAIN_DYN_LDS, s0, a);
    CZ_HIP(c, hipGetLastError());
    if (!cap) CZ_HIP(c, hipEventRecord(c->ev_stop, s0));
    c->timed = true; c->last_grid = grid;
    return CZ_OK;
}

/* A replay of the captured launch: the same kernels, grids, arguments and dependencies, submitted as one graph.  The start and stop
   events stay outside it, on the caller's stream, so cz_context_last_kernel_ms covers the whole graph. */
static int cz_replay(cz_context* c) {
    CZ_HIP(c, hipEventRecord(c->ev_start, c->stream));
    CZ_HIP(c, hipGraphLaunch(c->g_exec, c->stream));
    CZ_HIP(c, hipEventRecord(c->ev_stop, c->stream));
    c->timed = true; c->timed_chain = c->g_timed_chain; c->timed_exec = c->g_timed_exec; c->timed_lit = c->g_timed_lit; c->timed_wx = c->g_timed_wx;
    c->timed_small = c->g_timed_small; c->last_grid = c->g_grid; c->g_replayed = true;
    return CZ_OK;
}
/* Captures the launch on a stream of the context's own (cz_enqueue does exactly what it does otherwise) and replays it at once.
   CZ_E_NOGRAPH: nothing was enqueued — the caller goes the ordinary way, and this context does not try again. */
static int cz_capture(cz_context* c, const cz_batch_args& proto, size_t n) {
    int where = 0, rc = 0; hipError_t e = hipSuccess, e2 = hipSuccess;
    auto give_up = [&]() {
        if (getenv("CZ_GRAPH_DEBUG")) fprintf(stderr, "cz_capture: gave up at %d (enqueue rc %d, hip error of the context %d at line %d, end capture %d, instantiate %d, last %d)\n", where, rc, c->last_hip_error, c->last_hip_line, (int)e, (int)e2, (int)hipGetLastError());
        (void)hipGetLastError(); c->g_bad = true; return CZ_E_NOGRAPH; };
    if (!c->gstream && hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking) != hipSuccess) { c->gstream = nullptr; return give_up(); }
    if (!c->ev_lit_dep && hipEventCreateWithFlags(&c->ev_lit_dep, hipEventDisableTiming) != hipSuccess) { c->ev_lit_dep = nullptr; return give_up(); }
    if (!c->ev_small_dep && hipEventCreateWithFlags(&c->ev_small_dep, hipEventDisableTiming) != hipSuccess) { c->ev_small_dep = nullptr; return give_up(); }
    if (c->g_exec) { if (hipStreamSynchronize(c->stream) != hipSuccess) return give_up(); (void)hipGraphExecDestroy(c->g_exec); c->g_exec = nullptr; }   /* (its last replay may still be running) */
    where = 1;
    if (hipStreamBeginCapture(c->gstream, hipStreamCaptureModeRelaxed) != hipSuccess) return give_up();
    c->capturing = true;
    rc = cz_enqueue(c, proto, n, c->gstream);
    c->capturing = false;
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(c->gstream, &g);
    where = 2;
    if (rc != CZ_OK || e != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); return give_up(); }
    hipGraphExec_t ex = nullptr;
    e2 = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    where = 3;
    if (e2 != hipSuccess || !ex) return give_up();
    c->g_exec = ex; c->g_proto = proto; c->g_n = n; c->g_gen = c->cfg_gen;
    c->g_timed_chain = c->timed_chain; c->g_timed_exec = c->timed_exec; c->g_timed_lit = c->timed_lit; c->g_timed_wx = c->timed_wx; c->g_timed_small = c->timed_small;
    c->g_grid = c->last_grid;
    return cz_replay(c);
}
/* One batch launch.  With cz_context_set_graph_replay(ctx, 1) the second launch in a row with the same arguments and context settings
   is captured as a hipGraph; from then on such a launch is a replay.  What a launch enqueues depends on nothing
   but its arguments and the context (no device read-back), so a replay is the same work by construction; the BYTES behind the
   pointers may change between replays. */
static int cz_launch(cz_context* c, const cz_batch_args& proto, size_t n) {
    if (n == 0) return CZ_OK;
    if (n > 0xFFFFFFFFull) return CZ_E_INVALID_ARG;
    /* (not the early-execute arrangement: capturing its four streams ended in a host-side crash inside the ROCm 7.2 runtime) */
    const bool eligible = c->graph_mode && !c->g_bad && !proto.tasks && c->chain_arena && !c->early_execute;
    if (eligible) {
        if (c->g_exec && c->g_n == n && c->g_gen == c->cfg_gen && !memcmp(&c->g_proto, &proto, sizeof proto)) return cz_replay(c);
        if (c->g_seen && c->g_seen_n == n && c->g_seen_gen == c->cfg_gen && !memcmp(&c->g_seen_proto, &proto, sizeof proto)) {
            const int rc = cz_capture(c, proto, n);
            if (rc != CZ_E_NOGRAPH) return rc;
        }
    }
    c->g_replayed = false;
    const int rc = cz_enqueue(c, proto, n, c->stream);
    c->g_seen = eligible && rc == CZ_OK;
    if (c->g_seen) { c->g_seen_proto = proto; c->g_seen_n = n; c->g_seen_gen = c->cfg_gen; }
    return rc;
}

/* What the arenas of a batch must hold, from the headers alone (cz_scan_kernel's counting pass over the batch as it sits on the
   device; nothing is decoded): chain arena bytes = 8 per sequence + 1 312 per block with sequences + the reserved head, literal
   arena bytes = the Huffman-coded literals of blocks that have sequences + 16 per such block.  Synchronises; meant for set-up, not
   for the hot path. */
CZ_EXPORT int cz_context_measure_batch(cz_context* c, const void* d_in_base, const uint64_t* d_in_off, const uint64_t* d_in_len, size_t n, const uint64_t* d_out_cap,
                                       size_t* chain_arena_bytes, size_t* literal_arena_bytes) try {
    if (!c || (n && (!d_in_base || !d_in_off || !d_in_len || !d_out_cap))) return CZ_E_INVALID_ARG;
    if (chain_arena_bytes) *chain_arena_bytes = 0;
    if (literal_arena_bytes) *literal_arena_bytes = 0;
    if (!n) return CZ_OK;
    if (n > 0xFFFFFFFFull) return CZ_E_INVALID_ARG;
    CZ_HIP(c, hipSetDevice(c->device));
    const size_t waves = (n + CZ_WG_THREADS - 1) / CZ_WG_THREADS;
    std::vector<uint64_t> h(2 * n);                                     /* (before the device buffer: a bad_alloc here leaves nothing to free) */
    uint8_t* tmp = nullptr;                                             /* frame_first | lit_first | frame_pre | scan_ctl | scan_wave */
    const size_t o_ff = 0, o_lf = o_ff + n * 8, o_fp = o_lf + n * 8, o_ctl = (o_fp + n * 4 + 15) & ~(size_t)15, o_sw = o_ctl + CZ_SCAN_CTL_WORDS * 4, total = o_sw + waves * 72 * 4;
    CZ_HIP(c, hipMalloc((void**)&tmp, total));
    int st = CZ_OK;
    do {
        if (hipMemsetAsync(tmp + o_ctl, 0, CZ_SCAN_CTL_WORDS * 4, c->stream) != hipSuccess) { st = CZ_E_HIP; break; }
        cz_batch_args a; memset(&a, 0, sizeof a);
        a.in_base = (const uint8_t*)d_in_base; a.in_off = d_in_off; a.in_len = d_in_len; a.out_cap = d_out_cap; a.n = (uint32_t)n;
        a.frame_first = (uint64_t*)(tmp + o_ff); a.lit_first = (uint64_t*)(tmp + o_lf); a.frame_pre = (uint32_t*)(tmp + o_fp);
        a.scan_ctl = (uint32_t*)(tmp + o_ctl); a.scan_wave = (uint32_t*)(tmp + o_sw);
        a.lit_arena = tmp;                                              /* (not touched by the counting pass: it only says "count literals too") */
        a.chain_min_nseq = c->chain_min_nseq; a.scan_pass = 0;
        hipLaunchKernelGGL(cz_scan_kernel, dim3((unsigned)waves), dim3(CZ_WG_THREADS), 0, c->stream, a);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { st = CZ_E_HIP; break; }
        if (hipMemcpy(h.data(), tmp, 2 * n * 8, hipMemcpyDeviceToHost) != hipSuccess) { st = CZ_E_HIP; break; }
        unsigned long long units = 0, lbytes = 0;
        for (size_t i = 0; i < n; i++) { units += h[i]; lbytes += h[n + i]; }
        if (chain_arena_bytes) *chain_arena_bytes = (size_t)((units + 64 + 4096) * 8);
        if (literal_arena_bytes) *literal_arena_bytes = (size_t)(lbytes + 64 + 4096);
    } while (0);
    (void)hipFree(tmp);
    return st;
} catch (const std::bad_alloc&) { return CZ_E_OUT_OF_MEMORY; }

CZ_EXPORT int cz_decode_batch_device(cz_context* c, const void* d_in_base, const uint64_t* d_in_off, const uint64_t* d_in_len, size_t n,
                                     void* d_out_base, const uint64_t* d_out_off, const uint64_t* d_out_cap, cz_frame_result* d_results) {
    if (!c) return CZ_E_INVALID_ARG;
    if (n && (!d_in_base || !d_in_off || !d_in_len || !d_out_base || !d_out_off || !d_out_cap || !d_results)) return CZ_E_INVALID_ARG;
    CZ_HIP(c, hipSetDevice(c->device));
    cz_batch_args a; memset(&a, 0, sizeof a);
    a.in_base = (const uint8_t*)d_in_base; a.in_off = d_in_off; a.in_len = d_in_len;
    a.out_base = (uint8_t*)d_out_base; a.out_off = d_out_off; a.out_cap = d_out_cap; a.results = d_results; a.tasks = nullptr;
    return cz_launch(c, a, n);
}

static int cz_stage_reserve(cz_context* c, size_t bytes) {
    if (c->d_stage_bytes >= bytes) return CZ_OK;
    if (c->d_stage) { CZ_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_stage); c->d_stage = nullptr; c->d_stage_bytes = 0; }
    CZ_HIP(c, hipMalloc(&c->d_stage, bytes));
    c->d_stage_bytes = bytes;
    return CZ_OK;
}

CZ_EXPORT int cz_decode_batch_host(cz_context* c, const void* in_base, size_t in_bytes, const uint64_t* in_off, const uint64_t* in_len, size_t n,
                                   void* out_base, size_t out_bytes, const uint64_t* out_off, const uint64_t* out_cap, cz_frame_result* results) {
    if (!c) return CZ_E_INVALID_ARG;
    if (n == 0) return CZ_OK;
    if (!in_base || !in_off || !in_len || !out_base || !out_off || !out_cap || !results) return CZ_E_INVALID_ARG;
    for (size_t i = 0; i < n; i++) {
        if (in_off[i] > in_bytes || in_len[i] > in_bytes - in_off[i]) return CZ_E_INVALID_ARG;
        if (out_off[i] > out_bytes || out_cap[i] > out_bytes - out_off[i]) return CZ_E_INVALID_ARG;
    }
    CZ_HIP(c, hipSetDevice(c->device));
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_in = 0, o_out = o_in + up(in_bytes + 16), o_desc = o_out + up(out_bytes + 16), o_res = o_desc + up(4 * n * 8);
    const size_t total = o_res + up(n * sizeof(cz_frame_result));
    int st = cz_stage_reserve(c, total); if (st) return st;
    uint8_t* d = (uint8_t*)c->d_stage;
    uint64_t* d_desc = (uint64_t*)(d + o_desc);
    CZ_HIP(c, hipMemcpyAsync(d + o_in, in_base, in_bytes, hipMemcpyHostToDevice, c->stream));
    CZ_HIP(c, hipMemcpyAsync(d_desc, in_off, n * 8, hipMemcpyHostToDevice, c->stream));
    CZ_HIP(c, hipMemcpyAsync(d_desc + n, in_len, n * 8, hipMemcpyHostToDevice, c->stream));
    CZ_HIP(c, hipMemcpyAsync(d_desc + 2 * n, out_off, n * 8, hipMemcpyHostToDevice, c->stream));
    CZ_HIP(c, hipMemcpyAsync(d_desc + 3 * n, out_cap, n * 8, hipMemcpyHostToDevice, c->stream));
    st = cz_decode_batch_device(c, d + o_in, d_desc, d_desc + n, n, d + o_out, d_desc + 2 * n, d_desc + 3 * n, (cz_frame_result*)(d + o_res));
    if (st) return st;
    CZ_HIP(c, hipMemcpyAsync(out_base, d + o_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    CZ_HIP(c, hipMemcpyAsync(results, d + o_res, n * sizeof(cz_frame_result), hipMemcpyDeviceToHost, c->stream));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    return CZ_OK;
}

/* ------------------------------------------------------------------ several devices */
CZ_EXPORT int cz_partition_balanced(const uint64_t* weights, size_t n, size_t parts, uint32_t* part_of) try {
    if (!parts || (n && (!weights || !part_of))) return CZ_E_INVALID_ARG;
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weights[a] > weights[b]; });   /* heaviest first, index breaks ties */
    typedef std::pair<unsigned long long, size_t> load_t;                /* (load, part): the lightest part first, ties to the lower index */
    std::priority_queue<load_t, std::vector<load_t>, std::greater<load_t>> heap;
    for (size_t r = 0; r < parts; r++) heap.push(load_t(0ull, r));
    for (size_t k = 0; k < n; k++) {
        load_t t = heap.top(); heap.pop();
        part_of[order[k]] = (uint32_t)t.second;
        t.first += weights[order[k]];
        heap.push(t);
    }
    return CZ_OK;
} catch (const std::bad_alloc&) { return CZ_E_OUT_OF_MEMORY; }

/* (the staging buffer is only ever touched by cz_decode_batch_multi's worker of this context, which waits for the stream — in
   cz_decode_batch_host — before it returns: no copy is in flight when the buffer is grown here) */
static int cz_pin_reserve(cz_context* c, size_t bytes) {
    if (c->h_pin_bytes >= bytes) return CZ_OK;
    if (c->h_pin) { CZ_HIP(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->h_pin_bytes = 0; }
    CZ_HIP(c, hipHostMalloc(&c->h_pin, bytes, hipHostMallocDefault)); c->h_pin_bytes = bytes;
    return CZ_OK;
}
static int cz_distinct_contexts(cz_context* const* ctxs, size_t n_ctx) {
    if (!ctxs || !n_ctx) return 0;
    for (size_t d = 0; d < n_ctx; d++) { if (!ctxs[d]) return 0; for (size_t e = 0; e < d; e++) if (ctxs[e] == ctxs[d]) return 0; }   /* a context twice would race on its stream and staging */
    return 1;
}
CZ_EXPORT int cz_decode_batch_multi(cz_context* const* ctxs, size_t n_ctx, const void* in_base, size_t in_bytes, const uint64_t* in_off, const uint64_t* in_len, size_t n,
                                    void* out_base, size_t out_bytes, const uint64_t* out_off, const uint64_t* out_cap, cz_frame_result* results, uint32_t* device_of) try {
    if (!cz_distinct_contexts(ctxs, n_ctx)) return CZ_E_INVALID_ARG;
    if (n == 0) return CZ_OK;
    if (!in_base || !in_off || !in_len || !out_base || !out_off || !out_cap || !results) return CZ_E_INVALID_ARG;
    for (size_t i = 0; i < n; i++) {
        if (in_off[i] > in_bytes || in_len[i] > in_bytes - in_off[i]) return CZ_E_INVALID_ARG;
        if (out_off[i] > out_bytes || out_cap[i] > out_bytes - out_off[i]) return CZ_E_INVALID_ARG;
    }
    std::vector<uint64_t> weight(n);
    for (size_t i = 0; i < n; i++) weight[i] = in_len[i] + out_cap[i];
    std::vector<uint32_t> part(n);
    int st = cz_partition_balanced(weight.data(), n, n_ctx, part.data());
    if (st) return st;
    if (device_of) memcpy(device_of, part.data(), n * sizeof(uint32_t));
    std::vector<int> status(n_ctx, CZ_OK);
    std::vector<std::thread> workers;
    workers.reserve(n_ctx);
    /* (a thread that cannot be started must not unwind past the ones already running: joinable std::thread objects would end the
       process in their destructors — they are joined first, then the caller gets CZ_E_OUT_OF_MEMORY) */
    struct Joiner { std::vector<std::thread>& w; ~Joiner() { for (auto& t : w) if (t.joinable()) t.join(); } } joiner{workers};
    for (size_t d = 0; d < n_ctx; d++) {
        workers.emplace_back([&, d]() {
            try {
                /* this device's share, packed straight into the context's PINNED staging buffer (one copy on the host, and the
                   transfers to and from the device run at the bus's rate, asynchronously): descriptors, compressed frames back to
                   back (16-byte aligned), results, outputs in a layout of its own */
                std::vector<size_t> mine;
                for (size_t i = 0; i < n; i++) if (part[i] == d) mine.push_back(i);
                if (mine.empty()) return;
                const size_t m = mine.size();
                size_t ib = 0, ob = 0;
                for (size_t k = 0; k < m; k++) { ib += (size_t)((in_len[mine[k]] + 15) & ~15ull); ob += (size_t)((out_cap[mine[k]] + 255) & ~255ull); }
                auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
                const size_t p_desc = 0, p_res = p_desc + up(4 * m * 8), p_in = p_res + up(m * sizeof(cz_frame_result)), p_out = p_in + up(ib + 16), total = p_out + up(ob + 16);
                cz_context* c = ctxs[d];
                if (hipSetDevice(c->device) != hipSuccess) { status[d] = CZ_E_HIP; return; }
                status[d] = cz_pin_reserve(c, total); if (status[d]) return;
                uint8_t* hp = (uint8_t*)c->h_pin;
                uint64_t* ioff = (uint64_t*)(hp + p_desc), *ilen = ioff + m, *ooff = ilen + m, *ocap = ooff + m;
                size_t ia = 0, oa = 0;
                for (size_t k = 0; k < m; k++) {
                    ioff[k] = ia; ilen[k] = in_len[mine[k]]; ia += (size_t)((in_len[mine[k]] + 15) & ~15ull);
                    ooff[k] = oa; ocap[k] = out_cap[mine[k]]; oa += (size_t)((out_cap[mine[k]] + 255) & ~255ull);
                    memcpy(hp + p_in + ioff[k], (const uint8_t*)in_base + in_off[mine[k]], (size_t)ilen[k]);
                }
                cz_frame_result* res = (cz_frame_result*)(hp + p_res);
                status[d] = cz_decode_batch_host(c, hp + p_in, ib, ioff, ilen, m, hp + p_out, ob, ooff, ocap, res);
                if (status[d]) return;
                for (size_t k = 0; k < m; k++) {
                    results[mine[k]] = res[k];
                    const uint64_t w = res[k].bytes_produced < ocap[k] ? res[k].bytes_produced : ocap[k];
                    memcpy((uint8_t*)out_base + out_off[mine[k]], hp + p_out + ooff[k], (size_t)w);
                }
            } catch (const std::bad_alloc&) { status[d] = CZ_E_OUT_OF_MEMORY; }
        });
    }
    for (auto& t : workers) t.join();
    for (size_t d = 0; d < n_ctx; d++) if (status[d]) return status[d];
    return CZ_OK;
} catch (const std::bad_alloc&) { return CZ_E_OUT_OF_MEMORY; } catch (const std::system_error&) { return CZ_E_OUT_OF_MEMORY; }

/* The same with everything already on the devices — no host buffer, no PCIe in the path: share d (device pointers of context d's
   device) is launched on context d; the calls only enqueue (cz_decode_batch_device), so the devices run concurrently. */
CZ_EXPORT int cz_decode_batch_multi_device(cz_context* const* ctxs, size_t n_ctx, const cz_device_share* shares) {
    if (!cz_distinct_contexts(ctxs, n_ctx) || !shares) return CZ_E_INVALID_ARG;
    for (size_t d = 0; d < n_ctx; d++) {
        const cz_device_share& s = shares[d];
        if (!s.n) continue;
        const int st = cz_decode_batch_device(ctxs[d], s.d_in_base, s.d_in_off, s.d_in_len, s.n, s.d_out_base, s.d_out_off, s.d_out_cap, s.d_results);
        if (st) return st;
    }
    return CZ_OK;
}
/* The exchange step of SURVEY.md §8 (e): the decoded arenas of the other contexts to the root's device, as n_ctx - 1 CONCURRENT
   peer copies — each on its source context's stream, behind that context's decode — so that every xGMI link of the root carries
   one of them (a ring would be bound by one link).  The root's stream then waits for all of them: work enqueued on it afterwards
   (or cz_context_synchronize on the root) sees the gathered bytes. */
CZ_EXPORT int cz_gather_to_root(cz_context* const* ctxs, size_t n_ctx, size_t root, const void* const* d_src, const size_t* bytes, void* const* d_dst_on_root) {
    if (!cz_distinct_contexts(ctxs, n_ctx) || root >= n_ctx || !d_src || !bytes || !d_dst_on_root) return CZ_E_INVALID_ARG;
    cz_context* r = ctxs[root];
    for (size_t d = 0; d < n_ctx; d++) {
        if (d == root || !bytes[d]) continue;
        if (!d_src[d] || !d_dst_on_root[d]) return CZ_E_INVALID_ARG;
        cz_context* c = ctxs[d];
        CZ_HIP(c, hipSetDevice(c->device));
        CZ_HIP(c, hipMemcpyPeerAsync(d_dst_on_root[d], r->device, d_src[d], c->device, bytes[d], c->stream));
        hipEvent_t ev;
        CZ_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, c->stream);
        cz_context* at = c;
        if (e == hipSuccess) { at = r; e = hipSetDevice(r->device); }
        if (e == hipSuccess) e = hipStreamWaitEvent(r->stream, ev, 0);
        (void)hipEventDestroy(ev);                                      /* (released when it has completed; on every path) */
        if (e != hipSuccess) { at->last_hip_error = (int)e; return CZ_E_HIP; }
    }
    return CZ_OK;
}

/* ------------------------------------------------------------------ stateless parsers */
static inline uint32_t rd32le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

CZ_EXPORT int cz_read_frame_header(const uint8_t* p, size_t len, cz_frame_header* out, uint64_t* detail) {
    if (!out || (!p && len)) return CZ_E_INVALID_ARG;
    memset(out, 0, sizeof *out);
    if (len < 4) return CZ_E_FH_MAGIC_READ;                             /* frame.cairo:155-158 */
    const uint32_t magic = rd32le(p); size_t i = 4;
    if (magic >= 0x184D2A50u && magic <= 0x184D2A5Fu) {                 /* :160-166 */
        if (len < 8) return CZ_E_FH_DESCRIPTOR_READ;
        if (detail) { detail[0] = magic; detail[1] = rd32le(p + 4); }
        return CZ_E_FH_SKIP_FRAME;
    }
    if (magic != 0xFD2FB528u) { if (detail) detail[0] = magic; return CZ_E_FH_BAD_MAGIC; }   /* :168 */
    if (len < i + 1) return CZ_E_FH_DESCRIPTOR_READ;                    /* :172-175 */
    const uint8_t d = p[i++]; out->descriptor = d;
    const int single = (d >> 5) & 1;
    if (!single) { if (len < i + 1) return CZ_E_FH_WINDOW_DESC_READ; out->window_descriptor = p[i++]; }   /* :186-193 */
    const unsigned didf = d & 3, dl = didf == 3 ? 4 : didf;             /* :76-90 */
    if (dl) {                                                           /* :202-231 */
        if (len < i + dl) return CZ_E_FH_DICT_ID_READ;
        uint32_t id = 0; for (unsigned k = 0; k < dl; k++) id |= (uint32_t)p[i + k] << (8 * k);
        i += dl; if (id) { out->dict_id = id; out->has_dict_id = 1; }
    }
    const unsigned flag = d >> 6, fl = flag == 0 ? (single ? 1 : 0) : flag == 1 ? 2 : flag == 2 ? 4 : 8;  /* :56-74 */
    if (fl) {                                                           /* :240-278; truncation reports DictionaryIdReadError */
        if (len < i + fl) return CZ_E_FH_DICT_ID_READ;
        uint64_t f = 0; for (unsigned k = 0; k < fl; k++) f |= (uint64_t)p[i + k] << (8 * k);
        i += fl; if (fl == 2) f += 256;
        out->frame_content_size = f;
    }
    out->header_len = (uint8_t)i;
    if (single) out->window_size = out->frame_content_size;             /* :106-129 */
    else {
        const uint64_t base = 1ull << (10 + (out->window_descriptor >> 3)), w = base + (base / 8) * (out->window_descriptor & 7);
        if (w < 1024) return CZ_E_WINDOW_TOO_SMALL;
        if (w >= 4123168604160ull) return CZ_E_WINDOW_TOO_BIG;
        out->window_size = w;
    }
    return CZ_OK;
}

CZ_EXPORT int cz_read_block_header(const uint8_t* p, size_t len, cz_block_header* h) {
    if (!h || (!p && len)) return CZ_E_INVALID_ARG;
    memset(h, 0, sizeof *h);
    if (len < 3) return CZ_E_BH_TRUNCATED;                              /* (panic) block_decoder.cairo:240 */
    const uint32_t a = p[0], b = p[1], c = p[2], t = (a >> 1) & 3;      /* :289-304 */
    if (t == 3) return CZ_E_BH_RESERVED;                                /* :248 */
    const uint32_t size = (a >> 3) | (b << 5) | (c << 13);              /* :315-321 */
    if (size > 128u * 1024u) return CZ_E_BH_SIZE_TOO_LARGE;             /* :306-313 */
    h->block_type = (uint8_t)t; h->last_block = (uint8_t)(a & 1);
    h->decompressed_size = t == 2 ? 0 : size;                           /* :256-261 */
    h->content_size = t == 1 ? 1 : size;                                /* :262-267 */
    return CZ_OK;
}

/* ------------------------------------------------------------------ stream walker */
/* A .zst stream is a concatenation of zstd frames and skippable frames.  read_frame_header hands a skippable frame
 * back as the error SkipFrame(magic, length) for the CALLER to skip (src/frame.cairo:160-166); this is that caller:
 * it cuts the stream into batch entries — one per zstd frame, found by walking the frame's block headers
 * (block_decoder.cairo:237-321) to its last block and optional checksum — and steps over skippable frames
 * (8 bytes + length).  Host side, no decoding: the entries go to cz_decode_batch_* in ONE launch. */
CZ_EXPORT int cz_stream_split(const uint8_t* src, size_t len, cz_stream_entry* entries, size_t cap, size_t* count, size_t* consumed) {
    if ((!src && len) || !count || (!entries && cap)) return CZ_E_INVALID_ARG;
    size_t pos = 0, n = 0; int st = CZ_OK;
    while (pos < len) {
        cz_frame_header fh; uint64_t detail[2] = {0, 0};
        const int e = cz_read_frame_header(src + pos, len - pos, &fh, detail);
        cz_stream_entry ent; memset(&ent, 0, sizeof ent);
        ent.offset = pos;
        if (e == CZ_E_FH_SKIP_FRAME) {                                  /* frame.cairo:160-166 */
            if (len - pos - 8 < detail[1]) { st = CZ_E_BLOCK_TRUNCATED; break; }
            ent.kind = CZ_STREAM_SKIPPABLE; ent.magic = (uint32_t)detail[0]; ent.length = 8 + detail[1];
        } else if (e) { st = e; break; }
        else {
            size_t p = pos + fh.header_len; uint64_t bound = 0; int bad = 0;
            for (;;) {
                cz_block_header bh;
                const int be = cz_read_block_header(src + p, len - p, &bh);
                if (be) { bad = be; break; }
                if (len - p - 3 < bh.content_size) { bad = CZ_E_BLOCK_TRUNCATED; break; }
                bound += bh.block_type == 2 ? 128u * 1024u : bh.decompressed_size;
                p += 3 + bh.content_size;
                if (bh.last_block) break;
            }
            if (bad) { st = bad; break; }
            if ((fh.descriptor >> 2) & 1) { if (len - p < 4) { st = CZ_E_CHECKSUM_TRUNCATED; break; } p += 4; }
            ent.kind = CZ_STREAM_FRAME; ent.length = p - pos; ent.content_size = fh.frame_content_size; ent.out_bound = bound;
            ent.window_size = fh.window_size;
        }
        if (n < cap) entries[n] = ent;
        n++; pos += ent.length;
    }
    *count = n;
    if (consumed) *consumed = pos;
    if (!st && n > cap) return CZ_E_TARGET_TOO_SMALL;                   /* call again with room for *count entries */
    return st;
}

/* ------------------------------------------------------------------ XXH64 (frame content checksum) */
/* src/utils/xxhash64.cairo:20-163.  The reference hashes what DecodeBuffer drains
 * (decode_buffer.cairo:162,181), seed 0; the low 32 bits are compared (frame_decoder.cairo:133-138). */
namespace {
const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint64_t rnd(uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; }
inline uint64_t mrg(uint64_t h, uint64_t v) { v = rnd(0, v); h ^= v; return h * P1 + P4; }
struct Xxh64 {
    uint64_t total = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0; uint8_t mem[32]; uint32_t memsize = 0;
    void reset() { total = 0; memsize = 0; v1 = P1 + P2; v2 = P2; v3 = 0; v4 = 0 - P1; }
    void update(const uint8_t* p, size_t len) {
        total += len;
        if (memsize + len < 32) { memcpy(mem + memsize, p, len); memsize += (uint32_t)len; return; }
        const uint8_t* end = p + len;
        if (memsize) {
            size_t fill = 32 - memsize; memcpy(mem + memsize, p, fill);
            v1 = rnd(v1, rd64(mem)); v2 = rnd(v2, rd64(mem + 8)); v3 = rnd(v3, rd64(mem + 16)); v4 = rnd(v4, rd64(mem + 24));
            p += fill; memsize = 0;
        }
        while (p + 32 <= end) { v1 = rnd(v1, rd64(p)); v2 = rnd(v2, rd64(p + 8)); v3 = rnd(v3, rd64(p + 16)); v4 = rnd(v4, rd64(p + 24)); p += 32; }
        if (p < end) { memcpy(mem, p, (size_t)(end - p)); memsize = (uint32_t)(end - p); }
    }
    uint64_t digest() const {
        uint64_t h;
        if (total >= 32) { h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18); h = mrg(h, v1); h = mrg(h, v2); h = mrg(h, v3); h = mrg(h, v4); }
        else h = P5;
        h += total;
        const uint8_t* p = mem; const uint8_t* end = p + memsize;
        while (p + 8 <= end) { h ^= rnd(0, rd64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
        if (p + 4 <= end) { h ^= (uint64_t)rd32le(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
        while (p < end) { h ^= (uint64_t)(*p) * P5; h = rotl(h, 11) * P1; p++; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        return h;
    }
};
}  // namespace

/* ------------------------------------------------------------------ DecoderScratch on the device */
/* DecoderScratch (src/decoding/scratch.cairo:11-67): the per-frame carried state (Huffman table, three FSE tables,
 * RLE symbols, offset history: cz_device_frame_state in HBM) and the DecodeBuffer (decode_buffer.cairo:9-15).  The
 * decoded frame stays in HBM; `drained` marks how much the host already collected (buffer.len() == produced - drained)
 * and `base_off` how many leading frame bytes were dropped from the resident buffer (only drained bytes are ever
 * dropped, and a match can only reach bytes that are still in the buffer, decode_buffer.cairo:65). */
struct cz_decoder_scratch {
    cz_context* ctx = nullptr;
    uint64_t window_size = 0;
    uint8_t* d_out = nullptr; size_t d_out_cap = 0; uint64_t base_off = 0; uint64_t produced = 0, drained = 0;
    Xxh64 hash;
    uint8_t* d_src = nullptr; size_t d_src_cap = 0;
    uint8_t* d_ctl = nullptr;   /* [state | state backup | task | result] */
    const struct cz_dictionary* dict = nullptr;   /* DecodeBuffer.dict_content comes from it (init_from_dict); cleared by reset */
};
static const size_t CTL_STATE = 0, CTL_BACKUP = (sizeof(cz_device_frame_state) + 255) & ~(size_t)255,
                    CTL_TASK = 2 * CTL_BACKUP, CTL_RES = CTL_TASK + 256, CTL_BYTES = CTL_RES + 256;

static int scratch_reset_state(cz_decoder_scratch* s, uint64_t window_size) {      /* scratch.cairo:23-58 */
    cz_context* c = s->ctx;
    CZ_HIP(c, hipSetDevice(c->device));
    cz_device_frame_state init; memset(&init, 0, sizeof init);
    init.hist[0] = 1; init.hist[1] = 4; init.hist[2] = 8; init.fse_rle[0] = init.fse_rle[1] = init.fse_rle[2] = -1;
    CZ_HIP(c, hipMemcpyAsync(s->d_ctl + CTL_STATE, &init, sizeof init, hipMemcpyHostToDevice, c->stream));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    s->window_size = window_size; s->produced = 0; s->drained = 0; s->base_off = 0; s->hash.reset();
    s->dict = nullptr;                                                  /* decode_buffer.cairo:38 */
    return CZ_OK;
}
CZ_EXPORT int cz_decoder_scratch_create(cz_context* ctx, uint64_t window_size, cz_decoder_scratch** out) {
    if (!ctx || !out) return CZ_E_INVALID_ARG;
    *out = nullptr;
    cz_decoder_scratch* s = new (std::nothrow) cz_decoder_scratch();
    if (!s) return CZ_E_INVALID_ARG;
    s->ctx = ctx;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&s->d_ctl, CTL_BYTES) != hipSuccess) { delete s; return CZ_E_HIP; }
    const int st = scratch_reset_state(s, window_size);
    if (st) { (void)hipFree(s->d_ctl); delete s; return st; }
    *out = s; return CZ_OK;
}
CZ_EXPORT int cz_decoder_scratch_reset(cz_decoder_scratch* s, uint64_t window_size) { return s ? scratch_reset_state(s, window_size) : CZ_E_INVALID_ARG; }
CZ_EXPORT void cz_decoder_scratch_destroy(cz_decoder_scratch* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->d_out) (void)hipFree(s->d_out);
    if (s->d_src) (void)hipFree(s->d_src);
    if (s->d_ctl) (void)hipFree(s->d_ctl);
    delete s;
}
CZ_EXPORT size_t cz_decoder_scratch_buffer_len(const cz_decoder_scratch* s) { return s ? (size_t)(s->produced - s->drained) : 0; }   /* buffer.len() */
CZ_EXPORT uint64_t cz_decoder_scratch_total_output(const cz_decoder_scratch* s) { return s ? s->produced : 0; }              /* bytes decoded so far (the reference's total_output_counter lags behind it after whole-dictionary matches: see the header) */

/* DictionaryTrait::decode_dict (dictionary.cairo:35-91) */
CZ_EXPORT int cz_dictionary_decode(cz_context* c, const uint8_t* raw, size_t len, cz_dictionary** out, uint64_t* detail) {
    if (!c || !out || (!raw && len)) return CZ_E_INVALID_ARG;
    *out = nullptr;
    if (detail) { detail[0] = 0; detail[1] = 0; }
    CZ_HIP(c, hipSetDevice(c->device));
    cz_dictionary* d = new (std::nothrow) cz_dictionary();
    if (!d) return CZ_E_INVALID_ARG;
    d->ctx = c; d->len = len;
    uint64_t* d_res = nullptr;
    auto fail = [&](int st) { if (d->d_raw) (void)hipFree(d->d_raw); if (d->d_state) (void)hipFree(d->d_state); if (d_res) (void)hipFree(d_res); delete d; return st; };
    if (hipMalloc((void**)&d->d_raw, len + 16) != hipSuccess || hipMalloc((void**)&d->d_state, sizeof(cz_device_frame_state)) != hipSuccess ||
        hipMalloc((void**)&d_res, 64) != hipSuccess) return fail(CZ_E_HIP);
    if (len && hipMemcpyAsync(d->d_raw, raw, len, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(CZ_E_HIP);
    hipLaunchKernelGGL(cz_dict_setup_kernel, dim3(1), dim3(CZ_WG_THREADS), CZ_FSE_LDS_BYTES, c->stream, (const uint8_t*)d->d_raw, (uint64_t)len, d->d_state, d_res);
    uint64_t res[4] = {0, 0, 0, 0};
    cz_device_frame_state* hs = new (std::nothrow) cz_device_frame_state();
    if (!hs || hipGetLastError() != hipSuccess || hipMemcpyAsync(res, d_res, sizeof res, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { delete hs; return fail(CZ_E_HIP); }
    if (detail) detail[0] = res[3];                                     /* the magic number read (BadMagicNum) */
    if (res[0]) { delete hs; return fail((int)res[0]); }
    if (hipMemcpy(hs, d->d_state, sizeof *hs, hipMemcpyDeviceToHost) != hipSuccess) { delete hs; return fail(CZ_E_HIP); }
    d->content_off = (size_t)res[1]; d->id = (uint32_t)res[2];
    for (int k = 0; k < 3; k++) d->hist[k] = hs->hist[k];
    delete hs;
    (void)hipFree(d_res);
    *out = d;
    return CZ_OK;
}
CZ_EXPORT void cz_dictionary_destroy(cz_dictionary* d) {
    if (!d) return;
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->d_raw) (void)hipFree(d->d_raw);
    if (d->d_state) (void)hipFree(d->d_state);
    delete d;
}
/* Batch decodes of this context start every frame from `d` (NULL: from nothing, the default). */
CZ_EXPORT int cz_context_set_dictionary(cz_context* c, const cz_dictionary* d) {
    if (!c || (d && d->ctx != c)) return CZ_E_INVALID_ARG;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    c->cfg_gen++;
    c->batch_dict = d;
    return CZ_OK;
}
CZ_EXPORT uint32_t cz_dictionary_id(const cz_dictionary* d) { return d ? d->id : 0; }
CZ_EXPORT size_t cz_dictionary_content_len(const cz_dictionary* d) { return d ? d->len - d->content_off : 0; }
CZ_EXPORT int cz_dictionary_offset_hist(const cz_dictionary* d, uint32_t out[3]) {
    if (!d || !out) return CZ_E_INVALID_ARG;
    for (int k = 0; k < 3; k++) out[k] = d->hist[k];
    return CZ_OK;
}
/* DecoderScratchTrait::init_from_dict (scratch.cairo:60-65): tables, repeat offsets and dict_content of the workspace
   become the dictionary's.  The dictionary must outlive the workspace's use of it. */
CZ_EXPORT int cz_decoder_scratch_init_from_dict(cz_decoder_scratch* s, const cz_dictionary* d) {
    if (!s || !d || s->ctx != d->ctx) return CZ_E_INVALID_ARG;
    cz_context* c = s->ctx;
    CZ_HIP(c, hipSetDevice(c->device));
    CZ_HIP(c, hipMemcpyAsync(s->d_ctl + CTL_STATE, d->d_state, sizeof(cz_device_frame_state), hipMemcpyDeviceToDevice, c->stream));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    s->dict = d;
    return CZ_OK;
}

/* Makes room for `need` resident bytes behind base_off, keeping what a match or a drain can still reach:
 * first drops the drained prefix (compaction), then grows. */
static int scratch_reserve_out(cz_decoder_scratch* s, size_t need_abs /* frame position the buffer must reach */) {
    cz_context* c = s->ctx;
    if (need_abs - s->base_off <= s->d_out_cap) return CZ_OK;
    /* a fresh buffer for [drained, need_abs): the drained prefix is gone for the reference too */
    const size_t live = (size_t)(s->produced - s->drained), need = (size_t)(need_abs - s->drained);
    size_t nc = s->d_out_cap ? s->d_out_cap : (size_t)1 << 20;
    while (nc < need) nc *= 2;
    uint8_t* p = nullptr;
    CZ_HIP(c, hipMalloc((void**)&p, nc));
    if (s->d_out && live) CZ_HIP(c, hipMemcpyAsync(p, s->d_out + (s->drained - s->base_off), live, hipMemcpyDeviceToDevice, c->stream));
    CZ_HIP(c, hipStreamSynchronize(c->stream));
    if (s->d_out) (void)hipFree(s->d_out);
    s->d_out = p; s->d_out_cap = nc; s->base_off = s->drained;
    return CZ_OK;
}

/* Uploads `src` (positioned at a block header), runs the kernel on ONE frame task, returns the device's result record. */
static int scratch_run(cz_decoder_scratch* s, const uint8_t* src, size_t len, uint32_t strategy, uint64_t n, uint32_t streaming,
                       uint32_t has_checksum, cz_frame_result* res) {
    cz_context* c = s->ctx;
    CZ_HIP(c, hipSetDevice(c->device));
    if (s->d_src_cap < len + 16) {
        if (s->d_src) (void)hipFree(s->d_src);
        s->d_src = nullptr; s->d_src_cap = 0;
        size_t nc = (len + 16 + 65535) & ~(size_t)65535;
        CZ_HIP(c, hipMalloc((void**)&s->d_src, nc)); s->d_src_cap = nc;
    }
    if (len) CZ_HIP(c, hipMemcpyAsync(s->d_src, src, len, hipMemcpyHostToDevice, c->stream));
    /* output bound: a Raw / RLE block regenerates exactly its size; a compressed block at most 128 KiB in valid
       data — the reference does not enforce that, so grow and retry on CZ_E_OUTPUT_TOO_SMALL */
    size_t bound = 0, blocks = 0, pos = 0;
    while (len - pos >= 3) {
        cz_block_header bh; if (cz_read_block_header(src + pos, len - pos, &bh)) break;
        if (pos + 3 + bh.content_size > len) { if (!streaming) bound += bh.block_type == 2 ? 128u * 1024u : bh.decompressed_size; break; }
        bound += bh.block_type == 2 ? 128u * 1024u : bh.decompressed_size;
        blocks++; pos += 3 + bh.content_size;
        if (bh.last_block) break;
        if (strategy == CZ_STRATEGY_UPTO_BLOCKS && blocks >= n) break;
        if (strategy == CZ_STRATEGY_UPTO_BYTES && bound >= n + 128u * 1024u) break;
    }
    uint64_t want = s->produced + bound + 4096;
    CZ_HIP(c, hipMemcpyAsync(s->d_ctl + CTL_BACKUP, s->d_ctl + CTL_STATE, sizeof(cz_device_frame_state), hipMemcpyDeviceToDevice, c->stream));
    for (int attempt = 0; attempt < 8; attempt++) {
        int st = scratch_reserve_out(s, (size_t)want); if (st) return st;
        cz_device_task t; memset(&t, 0, sizeof t);
        t.src = s->d_src; t.src_len = len;
        t.dst = s->d_out - s->base_off;                                 /* frame position p lives at dst + p (p >= base_off) */
        t.dst_cap = s->base_off + s->d_out_cap; t.produced = s->produced; t.drained = s->drained;
        t.window_size = s->window_size; t.strategy = strategy; t.strategy_n = n; t.has_checksum = has_checksum; t.streaming = streaming;
        t.state = (cz_device_frame_state*)(s->d_ctl + CTL_STATE);
        if (s->dict) { t.dict = s->dict->d_raw + s->dict->content_off; t.dict_len = s->dict->len - s->dict->content_off; }
        CZ_HIP(c, hipMemcpyAsync(s->d_ctl + CTL_TASK, &t, sizeof t, hipMemcpyHostToDevice, c->stream));
        cz_batch_args a; memset(&a, 0, sizeof a);
        a.tasks = (const cz_device_task*)(s->d_ctl + CTL_TASK); a.results = (cz_frame_result*)(s->d_ctl + CTL_RES);
        st = cz_launch(c, a, 1); if (st) return st;
        CZ_HIP(c, hipMemcpyAsync(res, s->d_ctl + CTL_RES, sizeof *res, hipMemcpyDeviceToHost, c->stream));
        CZ_HIP(c, hipStreamSynchronize(c->stream));
        if (res->status != CZ_E_OUTPUT_TOO_SMALL) return CZ_OK;
        /* roll the carried state back and retry with a larger resident buffer */
        CZ_HIP(c, hipMemcpyAsync(s->d_ctl + CTL_STATE, s->d_ctl + CTL_BACKUP, sizeof(cz_device_frame_state), hipMemcpyDeviceToDevice, c->stream));
        want = s->drained + (uint64_t)(s->d_out_cap ? s->d_out_cap : (size_t)1 << 20) * 4;
    }
    return CZ_OK;
}
/* drain_to (decode_buffer.cairo:168-186): device -> host, hash update, advance */
static size_t scratch_drain(cz_decoder_scratch* s, size_t amount, uint8_t* dst, size_t cap) {
    const size_t bl = (size_t)(s->produced - s->drained);
    size_t n = bl < amount ? bl : amount;
    if (n > cap) n = cap;
    if (!n) return 0;
    cz_context* c = s->ctx;
    if (hipSetDevice(c->device) != hipSuccess) return 0;
    if (hipMemcpyAsync(dst, s->d_out + (s->drained - s->base_off), n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return 0;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    s->hash.update(dst, n);
    s->drained += n;
    return n;
}
/* DecodeBuffer::drain (decode_buffer.cairo:157-166): everything in the buffer; *written = 0 and CZ_E_TARGET_TOO_SMALL when dst cannot hold it. */
CZ_EXPORT int cz_decoder_scratch_drain(cz_decoder_scratch* s, uint8_t* dst, size_t cap, size_t* written) {
    if (!s || !written) return CZ_E_INVALID_ARG;
    *written = 0;
    const size_t bl = (size_t)(s->produced - s->drained);
    if (bl > cap) return CZ_E_TARGET_TOO_SMALL;
    *written = scratch_drain(s, bl, dst, cap);
    return CZ_OK;
}
/* DecodeBuffer::drain_to_window_size (:145-155): what exceeds window_size; returns 1 = Some, 0 = None, < 0 = -status. */
CZ_EXPORT int cz_decoder_scratch_drain_to_window_size(cz_decoder_scratch* s, uint8_t* dst, size_t cap, size_t* written) {
    if (!s || !written) return -CZ_E_INVALID_ARG;
    *written = 0;
    const size_t bl = (size_t)(s->produced - s->drained);
    if (bl <= s->window_size) return 0;
    const size_t can = (size_t)(bl - s->window_size);
    if (can > cap) return -CZ_E_TARGET_TOO_SMALL;
    *written = scratch_drain(s, can, dst, cap);
    return 1;
}
CZ_EXPORT uint64_t cz_decoder_scratch_hash_digest(const cz_decoder_scratch* s) { return s ? s->hash.digest() : 0; }   /* XXH64 of what was drained */

/* ------------------------------------------------------------------ block decoder */
/* BlockDecoder (src/decoding/block_decoder.cairo:20-30, :69-137, :237-278): the two-state machine around one block.
 * The body of a block is always decoded by the device kernel, against the DecoderScratch the caller passes. */
CZ_EXPORT void cz_block_decoder_new(cz_block_decoder* bd) { if (bd) { bd->internal_state = CZ_BLOCK_READY_FOR_HEADER; bd->header_buffer[0] = bd->header_buffer[1] = bd->header_buffer[2] = 0; } }
CZ_EXPORT int cz_block_decoder_read_block_header(cz_block_decoder* bd, const uint8_t* src, size_t len, cz_block_header* out, uint8_t* consumed) {
    if (!bd || !out) return CZ_E_INVALID_ARG;
    if (consumed) *consumed = 0;
    if (len < 3 || !src) return CZ_E_BH_TRUNCATED;                      /* (panic) r.slice(0, 3) :240 */
    bd->header_buffer[0] = src[0]; bd->header_buffer[1] = src[1]; bd->header_buffer[2] = src[2];      /* :241 */
    if (consumed) *consumed = 3;                                        /* the slice is advanced before the checks (:242) */
    const int e = cz_read_block_header(src, len, out);
    if (e) return e;                                                    /* header_buffer keeps the bytes, the state does not move */
    bd->header_buffer[0] = bd->header_buffer[1] = bd->header_buffer[2] = 0;                            /* reset_buffer :270 */
    bd->internal_state = CZ_BLOCK_READY_FOR_BODY;                       /* :271 */
    return CZ_OK;
}
CZ_EXPORT int cz_block_decoder_decode_block_content(cz_block_decoder* bd, const cz_block_header* h, cz_decoder_scratch* ws,
                                                    const uint8_t* src, size_t len, uint64_t* consumed) {
    if (!bd || !h || !ws || (!src && len)) return CZ_E_INVALID_ARG;
    if (consumed) *consumed = 0;
    if (bd->internal_state == CZ_BLOCK_READY_FOR_HEADER) return CZ_E_BLOCK_EXPECTED_HEADER;     /* :86-88 */
    if (bd->internal_state == CZ_BLOCK_FAILED) return CZ_E_BLOCK_STATE_FAILED;                  /* :90-92 */
    if (h->block_type > 2) return CZ_E_BH_RESERVED;                                             /* ReservedBlock :134 */
    const uint32_t size = h->block_type == 1 ? h->decompressed_size : h->content_size;
    if (len < h->content_size) return CZ_E_BLOCK_TRUNCATED;                                     /* (panic) :98,105,145 */
    /* the device task walks blocks from their header: put this block's header back in front of its content */
    std::vector<uint8_t> buf(3 + (size_t)h->content_size);
    const uint32_t v = (uint32_t)(h->last_block ? 1 : 0) | ((uint32_t)h->block_type << 1) | (size << 3);
    buf[0] = (uint8_t)v; buf[1] = (uint8_t)(v >> 8); buf[2] = (uint8_t)(v >> 16);
    if (h->content_size) memcpy(buf.data() + 3, src, h->content_size);
    cz_frame_result r; memset(&r, 0, sizeof r);
    const int st = scratch_run(ws, buf.data(), buf.size(), CZ_STRATEGY_UPTO_BLOCKS, 1, 0, 0, &r);
    if (st) return st;
    if (r.status) return r.status;                                      /* DecompressBlockError leaves; the state stays ReadyToDecodeNextBody like the reference */
    ws->produced = r.bytes_produced;
    bd->internal_state = CZ_BLOCK_READY_FOR_HEADER;                     /* :101,108,131 */
    if (consumed) *consumed = h->content_size;                          /* :102,122,132 */
    return CZ_OK;
}

/* ------------------------------------------------------------------ frame decoder */
struct cz_frame_decoder {
    cz_context* ctx = nullptr;
    cz_frame_header fh{}; bool initialised = false;
    bool frame_finished = false; size_t block_counter = 0; uint64_t bytes_read_counter = 0;    /* frame_decoder.cairo:22-30 */
    uint32_t check_sum = 0; bool has_check_sum = false;
    cz_decoder_scratch* scratch = nullptr;                              /* decoder_scratch (frame_decoder.cairo:25) */
};

CZ_EXPORT int cz_frame_decoder_create(cz_context* ctx, cz_frame_decoder** out) {
    if (!ctx || !out) return CZ_E_INVALID_ARG;
    *out = nullptr;
    cz_frame_decoder* fd = new (std::nothrow) cz_frame_decoder();
    if (!fd) return CZ_E_INVALID_ARG;
    fd->ctx = ctx;
    const int st = cz_decoder_scratch_create(ctx, 0, &fd->scratch);
    if (st) { delete fd; return st; }
    *out = fd; return CZ_OK;
}
CZ_EXPORT void cz_frame_decoder_destroy(cz_frame_decoder* fd) {
    if (!fd) return;
    cz_decoder_scratch_destroy(fd->scratch);
    delete fd;
}

static int fd_init(cz_frame_decoder* fd, const uint8_t* src, size_t len, size_t* consumed, uint64_t* detail, bool is_reset) {
    if (!fd) return CZ_E_INVALID_ARG;
    cz_frame_header fh;
    int e = cz_read_frame_header(src, len, &fh, detail);                /* frame_decoder.cairo:55-64 / :81-90 */
    if (e) return e;
    if (is_reset && fh.window_size > 1024ull * 1024 * 100) return CZ_E_WINDOW_SIZE_TOO_BIG;     /* :92 (D4: new() has no cap) */
    e = scratch_reset_state(fd->scratch, fh.window_size);              /* DecoderScratch::new / reset (scratch.cairo:23-58) */
    if (e) return e;
    fd->fh = fh; fd->initialised = true;
    fd->frame_finished = false; fd->block_counter = 0; fd->bytes_read_counter = fh.header_len;
    fd->check_sum = 0; fd->has_check_sum = false;
    if (consumed) *consumed = fh.header_len;
    return CZ_OK;
}
CZ_EXPORT int cz_frame_decoder_new(cz_frame_decoder* fd, const uint8_t* src, size_t len, size_t* consumed, uint64_t* detail) { return fd_init(fd, src, len, consumed, detail, false); }
CZ_EXPORT int cz_frame_decoder_reset(cz_frame_decoder* fd, const uint8_t* src, size_t len, size_t* consumed, uint64_t* detail) { return fd_init(fd, src, len, consumed, detail, true); }

CZ_EXPORT uint64_t cz_frame_decoder_content_size(const cz_frame_decoder* fd) { return fd ? fd->fh.frame_content_size : 0; }
CZ_EXPORT int cz_frame_decoder_checksum_from_data(const cz_frame_decoder* fd, uint32_t* v) { if (fd && fd->has_check_sum && v) *v = fd->check_sum; return fd && fd->has_check_sum; }
CZ_EXPORT uint32_t cz_frame_decoder_calculated_checksum(const cz_frame_decoder* fd) { return fd ? (uint32_t)fd->scratch->hash.digest() : 0; }
CZ_EXPORT uint64_t cz_frame_decoder_bytes_read_from_source(const cz_frame_decoder* fd) { return fd ? fd->bytes_read_counter : 0; }
CZ_EXPORT int cz_frame_decoder_is_finished(const cz_frame_decoder* fd) {
    if (!fd) return 0;
    if ((fd->fh.descriptor >> 2) & 1) return fd->frame_finished && fd->has_check_sum;           /* frame_decoder.cairo:144-150 */
    return fd->frame_finished;
}
CZ_EXPORT size_t cz_frame_decoder_blocks_decoded(const cz_frame_decoder* fd) { return fd ? fd->block_counter : 0; }
/* the DecoderScratch of this frame decoder (owned by it), e.g. to decode single blocks against it */
CZ_EXPORT cz_decoder_scratch* cz_frame_decoder_scratch(cz_frame_decoder* fd) { return fd ? fd->scratch : nullptr; }

static inline size_t fd_buffer_len(const cz_frame_decoder* fd) { return (size_t)(fd->scratch->produced - fd->scratch->drained); }

static void fd_fold(cz_frame_decoder* fd, const cz_frame_result& r) {
    fd->bytes_read_counter += r.bytes_consumed; fd->block_counter += r.blocks_decoded; fd->scratch->produced = r.bytes_produced;
    if (r.flags & CZ_RESULT_FINISHED) fd->frame_finished = true;
    if (r.flags & CZ_RESULT_HAS_CHECKSUM) { fd->check_sum = r.checksum_from_data; fd->has_check_sum = true; }
}

CZ_EXPORT int cz_frame_decoder_decode_blocks(cz_frame_decoder* fd, const uint8_t* src, size_t len, cz_strategy strategy, size_t n,
                                             size_t* consumed, int* finished) {
    if (!fd || !fd->initialised || (!src && len)) return CZ_E_INVALID_ARG;
    cz_frame_result r; memset(&r, 0, sizeof r);
    int st = scratch_run(fd->scratch, src, len, (uint32_t)strategy, n, 0, (fd->fh.descriptor >> 2) & 1, &r);       /* frame_decoder.cairo:156-222 */
    if (st) return st;
    fd_fold(fd, r);
    if (consumed) *consumed = (size_t)r.bytes_consumed;
    if (finished) *finished = fd->frame_finished;
    return r.status;
}

CZ_EXPORT size_t cz_frame_decoder_can_collect(const cz_frame_decoder* fd) {                    /* frame_decoder.cairo:233-243 */
    if (!fd) return 0;
    const size_t bl = fd_buffer_len(fd);
    if (cz_frame_decoder_is_finished(fd)) return bl;
    return bl > fd->scratch->window_size ? (size_t)(bl - fd->scratch->window_size) : 0;
}
CZ_EXPORT int cz_frame_decoder_collect(cz_frame_decoder* fd, uint8_t* dst, size_t cap, size_t* written) {     /* :224-231 */
    if (!fd || !written) return -CZ_E_INVALID_ARG;
    *written = 0;
    const size_t bl = fd_buffer_len(fd);
    if (cz_frame_decoder_is_finished(fd)) {
        if (bl > cap) return -CZ_E_TARGET_TOO_SMALL;
        *written = scratch_drain(fd->scratch, bl, dst, cap); return 1;
    }
    return cz_decoder_scratch_drain_to_window_size(fd->scratch, dst, cap, written);
}
CZ_EXPORT size_t cz_frame_decoder_read(cz_frame_decoder* fd, uint8_t* dst, size_t cap) {                     /* :328-334 */
    if (!fd) return 0;
    const size_t bl = fd_buffer_len(fd);
    const size_t amount = fd->frame_finished ? bl : (bl > fd->scratch->window_size ? (size_t)(bl - fd->scratch->window_size) : 0);
    return scratch_drain(fd->scratch, amount, dst, cap);
}
CZ_EXPORT int cz_frame_decoder_decode_from_to(cz_frame_decoder* fd, const uint8_t* src, size_t len, uint8_t* dst, size_t cap,
                                              size_t* read_len, size_t* written) {                           /* :245-326 */
    if (!fd || !fd->initialised || !read_len || !written || (!src && len)) return CZ_E_INVALID_ARG;
    const uint64_t start = fd->bytes_read_counter;
    *read_len = 0; *written = 0;
    if (!cz_frame_decoder_is_finished(fd)) {
        const bool cks = (fd->fh.descriptor >> 2) & 1;
        if (cks && fd->frame_finished && !fd->has_check_sum) {          /* :255-267 */
            if (len >= 4) { fd->check_sum = rd32le(src); fd->has_check_sum = true; fd->bytes_read_counter += 4; }
            *read_len = 4; return CZ_OK;                                /* (4, 0) even when fewer than 4 bytes were there */
        }
        cz_frame_result r; memset(&r, 0, sizeof r);
        int st = scratch_run(fd->scratch, src, len, CZ_STRATEGY_ALL, 0, 1, cks, &r);
        if (st) return st;
        fd_fold(fd, r);
        if (r.status) return r.status;
    }
    *written = cz_frame_decoder_read(fd, dst, cap);
    *read_len = (size_t)(fd->bytes_read_counter - start);
    return CZ_OK;
}
