/*
 * cairo_zstd_amd.h — C ABI of the MI355X-native zstd decoder (libcairo_zstd_amd.so).
 *
 * The reference (NethermindEth/cairo_zstd, pure Cairo 1) has no FFI of its own
 * (SURVEY.md §8b); the boundary is therefore its *trait API*, re-exposed here as plain C:
 * opaque handles, plain pointers and sizes, int32 status codes that map 1:1 onto the
 * reference's error-enum leaves (cairo_zstd_amd_status.h).  Every entry point cites the
 * reference interface it replaces (file:line relative to the reference tree).
 * INTEGRATION.md shows the Cairo-runner-hint / ctypes binding a maintainer would add.
 *
 * Three layers:
 *   1. cz_context_*        device context (HIP stream, scratch arenas).     [no reference analogue]
 *   2. cz_decode_batch*    many independent frames in one launch — the GPU hot path.
 *                          Semantics per frame = `_test_decode` (src/tests/decoding.cairo:4-21).
 *   3. cz_frame_decoder_*  one resumable frame, mirrors FrameDecoderTrait
 *                          (src/frame_decoder.cairo:107-335) call for call.
 *   plus stateless header parsers mirroring read_frame_header / read_block_header.
 *
 * Threading: one context / frame decoder per host thread.  Batch calls are asynchronous on
 * the context's HIP stream unless stated otherwise.  All compute runs on the device; there
 * is no CPU decode path in this library — without a gfx950 device every compute entry
 * returns CZ_E_NO_DEVICE.
 */
#ifndef CAIRO_ZSTD_AMD_H
#define CAIRO_ZSTD_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "cairo_zstd_amd_status.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CZ_ABI_VERSION 1

/* Per-frame result record written by the device (one per batch entry).  Mirrors what the
 * reference exposes after decode_blocks: is_finished (frame_decoder.cairo:144),
 * blocks_decoded (:152), bytes_read_from_source (:140), get_checksum_from_data (:129). */
typedef struct cz_frame_result {
    int32_t  status;            /* cz_status */
    uint32_t blocks_decoded;
    uint64_t bytes_consumed;    /* source bytes consumed, frame header and checksum included */
    uint64_t bytes_produced;    /* decoded bytes written at out_base + out_off[i] */
    uint32_t checksum_from_data;/* valid when flags & CZ_RESULT_HAS_CHECKSUM */
    uint32_t flags;
    uint64_t detail[2];         /* error payload (e.g. magic / skip length for CZ_E_FH_SKIP_FRAME,
                                   block index and byte position for decode errors) */
    uint32_t calculated_checksum; /* low 32 bits of XXH64(decoded frame), valid when flags & CZ_RESULT_CHECKSUM_COMPUTED
                                   (get_calculated_checksum, frame_decoder.cairo:133-138) */
    uint32_t reserved;
} cz_frame_result;
#define CZ_RESULT_FINISHED     1u   /* last block seen (frame_finished, frame_decoder.cairo:190) */
#define CZ_RESULT_HAS_CHECKSUM 2u
#define CZ_RESULT_CHECKSUM_COMPUTED 4u  /* cz_context_set_verify_checksum(ctx, 1): XXH64 was computed on the device */
#define CZ_RESULT_CHECKSUM_MATCH    8u  /* ... and equals checksum_from_data */

/* ---------------------------------------------------------------- 1. context */
typedef struct cz_context cz_context;

int  cz_abi_version(void);
/* Creates a decoder context on HIP device `device` (ordinal).  `stream` may be NULL (the
 * context then owns a NON-BLOCKING stream) or a hipStream_t cast to void*.  NULL is also the handle of the legacy default
 * stream: a caller whose other work runs there (PyTorch's torch.cuda.current_stream().cuda_stream is 0 unless a stream
 * context is active) gets a context that is NOT ordered with that work — it must synchronise (or record / wait on events)
 * between filling the buffers a batch call reads or writes and the call, and again before it reads the results, or pass a
 * stream of its own making. */
int  cz_context_create(cz_context** out, int device, void* stream);
void cz_context_destroy(cz_context* ctx);
int  cz_context_synchronize(cz_context* ctx);
/* Last HIP error seen by this context (hipError_t), for CZ_E_HIP diagnostics. */
int  cz_context_last_hip_error(const cz_context* ctx);
/* Kernel-launch geometry actually used (for bench / roofline reports). */
int  cz_context_launch_info(const cz_context* ctx, int* workgroups, int* threads_per_wg, int* compute_units);
/* Waves the execute-stage kernels of a batch launch run with: cz_execute_frames_kernel (4 waves per SIMD) and its 8-waves build. */
int  cz_context_execute_grid(const cz_context* ctx, int* waves, int* waves8);

/* ------------------------------------------------------- 2. batch (GPU hot path) */
/*
 * Decodes n independent zstd frames.  Frame i occupies in_base[in_off[i] .. +in_len[i]) and is
 * decoded to out_base[out_off[i] .. +out_cap[i]).  Each frame is handled exactly like the
 * reference's end-to-end entry: FrameDecoderStateTrait::new (frame_decoder.cairo:54) ->
 * decode_blocks(All) (:156) -> is_finished (:144); results[i] carries what the getters return.
 * A frame that fails leaves its neighbours untouched.
 *
 * ALL pointers are DEVICE pointers (results too).  Asynchronous on the context stream.
 * Returns CZ_OK when the launch was enqueued; per-frame outcomes are in results[].
 */
int cz_decode_batch_device(cz_context* ctx,
                           const void* d_in_base, const uint64_t* d_in_off, const uint64_t* d_in_len, size_t n,
                           void* d_out_base, const uint64_t* d_out_off, const uint64_t* d_out_cap,
                           cz_frame_result* d_results);

/* Same, with HOST buffers: stages the inputs to the device, decodes, copies outputs and
 * results back, and synchronizes.  (PCIe-inclusive convenience path.) */
int cz_decode_batch_host(cz_context* ctx,
                         const void* in_base, size_t in_bytes, const uint64_t* in_off, const uint64_t* in_len, size_t n,
                         void* out_base, size_t out_bytes, const uint64_t* out_off, const uint64_t* out_cap,
                         cz_frame_result* results);

/* Several devices: frames share nothing (FrameDecoderStateTrait::reset, src/frame_decoder.cairo:78-104),
 * so a batch shards by frame with no exchange between the devices.
 *
 * cz_partition_balanced deals n items of the given weights (a frame's algorithmic bytes: compressed + decoded) to `parts`
 * parts so that the sums are balanced — heaviest first, each to the part that is lightest so far, ties to the lower index:
 * deterministic, every caller computes the same dealing.  part_of[i] = part of item i.  No device needed.
 *
 * cz_decode_batch_multi decodes ONE batch of host buffers on n_ctx contexts (normally one per GPU of the node): it deals the
 * frames with cz_partition_balanced (weight = in_len + out_cap), runs every share through cz_decode_batch_host on a thread
 * of its own — stage, launch, copy back, all concurrently across the devices — and leaves outputs and results in the
 * caller's layout, exactly as one cz_decode_batch_host call would.  device_of (optional, n entries): the context index every
 * frame went to.  Every context keeps its own arenas and options (cz_context_set_chain_arena ... per context); a context may
 * appear once.  A share is packed once, into pinned staging memory the context keeps. */
int cz_partition_balanced(const uint64_t* weights, size_t n, size_t parts, uint32_t* part_of);
/* ... and without host buffers (no PCIe in the path): the caller has dealt the frames (cz_partition_balanced) and put every share
 * on its device; cz_decode_batch_multi_device launches share d on context d (cz_decode_batch_device: it only enqueues, so the
 * devices run concurrently).  A context may appear once.
 * cz_gather_to_root is the one exchange the path has: `bytes[d]` bytes at d_src[d] on context d's device go
 * to d_dst_on_root[d] on the root context's device, every d != root — n_ctx - 1 concurrent peer copies (hipMemcpyPeerAsync), each on
 * its source context's stream behind that context's decode, so that all of the root's xGMI links carry traffic; the root's stream
 * waits for them.  Entry `root` of the three arrays is ignored. */
typedef struct cz_device_share {
    const void* d_in_base; const uint64_t* d_in_off; const uint64_t* d_in_len; size_t n;
    void* d_out_base; const uint64_t* d_out_off; const uint64_t* d_out_cap; cz_frame_result* d_results;
} cz_device_share;
int cz_decode_batch_multi_device(cz_context* const* ctxs, size_t n_ctx, const cz_device_share* shares);
int cz_gather_to_root(cz_context* const* ctxs, size_t n_ctx, size_t root, const void* const* d_src, const size_t* bytes, void* const* d_dst_on_root);
int cz_decode_batch_multi(cz_context* const* ctxs, size_t n_ctx,
                          const void* in_base, size_t in_bytes, const uint64_t* in_off, const uint64_t* in_len, size_t n,
                          void* out_base, size_t out_bytes, const uint64_t* out_off, const uint64_t* out_cap,
                          cz_frame_result* results, uint32_t* device_of);

/* Enables (bytes > 0) or disables (0) the FSE-chain pre-pass for batch decodes on this context and sizes its record
 * arena (8 bytes per sequence + 1312 per block with sequences; 8x the compressed bytes + 64 MiB covers every BASELINE
 * config).  With the pre-pass a batch decode is: cz_scan_kernel twice (lists the blocks of all frames, sorted by
 * sequence count, and allocates their arena space) -> cz_chain_kernel (ten blocks per wave, the three FSE state
 * machines of a block on three lanes; writes one record per sequence) -> the frame kernels, which consume the
 * records (sequence_section_decoder.cairo:223-297 is what the records replace).  Frames the arena cannot hold, or
 * that are irregular in any way, are decoded entirely by cz_decode_frames_kernel: errors are only ever reported by it. */
int cz_context_set_chain_arena(cz_context* ctx, size_t bytes);
/* Enables (bytes > 0) or disables (0) the rest of the pre-pass, which needs the chain arena too: cz_scan_kernel also lists
 * every Huffman-coded literals section and every run of bytes whose place in the output is known without decoding
 * (Raw / RLE blocks ahead of a frame's first block with sequences); the sections are decoded block-parallel
 * (tree description, table, streams: literals_section_decoder.cairo:58-243) by cz_huf1_kernel NEXT TO cz_chain_kernel and
 * cz_huf_kernel behind it — into nodes of an arena of `bytes` (decoded literal bytes + 16 per block; at most the decoded
 * size of the batch), or straight into the output for blocks without sequences — and the runs are copied by
 * cz_tile_kernel.  Frames the pre-pass finishes need no frame kernel at all; the others go to cz_execute_frames_kernel
 * (sequence execution only: sequence_execution.cairo:12-129) unless cz_context_set_exec_kernel turned it off; frames that
 * did not fit, or that are irregular in any way, are decoded from scratch by cz_decode_frames_kernel in the same call. */
int cz_context_set_literal_arena(cz_context* ctx, size_t bytes);
/* What the two arenas must hold for a batch that already sits on the device, from its headers alone (cz_scan_kernel's counting pass:
 * block_decoder.cairo:237-321, literals_section.cairo:81-175, sequence_section.cairo:77-114; nothing is decoded): 8 bytes per
 * sequence + 1 312 per block with sequences, and the Huffman-coded literals of blocks with sequences + 16 per block.  For set-up
 * (it synchronises): size the arenas once for the largest batch a caller will decode — a rule of thumb (8 x the compressed bytes)
 * takes 2.5 x as much for config 4a. */
int cz_context_measure_batch(cz_context* ctx, const void* d_in_base, const uint64_t* d_in_off, const uint64_t* d_in_len, size_t n,
                             const uint64_t* d_out_cap, size_t* chain_arena_bytes, size_t* literal_arena_bytes);
/* Frames whose first sequences section holds fewer sequences than `n` skip the pre-pass (default 0: every frame
 * with sequences takes it — the pre-pass works block by block, so short chains cost little). */
int cz_context_set_chain_min_sequences(cz_context* ctx, uint32_t n);

/* Batch decodes also compute the XXH64 content checksum of every frame that carries one, on the
 * device, and compare it with the stored value (what `_test_decode` asserts,
 * src/tests/decoding.cairo:16-19).  A mismatch is reported in the result flags, not as a status:
 * the reference leaves the comparison to the caller too.  Off by default. */
int cz_context_set_verify_checksum(cz_context* ctx, int on);

/* Duration in milliseconds of the most recent decode launch on this context, measured with
 * hipEvents recorded on the context stream around the kernel (bench.py's roofline leg).
 * Blocks until that launch has finished. */
int cz_context_last_kernel_ms(cz_context* ctx, float* ms);
/* The part of it spent in the FSE-chain pre-pass kernel (0 when the pre-pass is off). */
int cz_context_last_chain_ms(cz_context* ctx, float* ms);
/* Diagnostics of the most recent batch launch of n frames (synchronises): frames that got chain records from the
 * pre-pass, frames whose literals the huff0 kernels decoded. */
int cz_context_last_prepass_counts(cz_context* ctx, size_t n, size_t* with_chain, size_t* with_literals);
/* How long that launch went on with cz_huf_kernel / cz_huf1_kernel / cz_tile_kernel after the chain kernel was done
 * (0: no literal arena). */
int cz_context_last_literals_tail_ms(cz_context* ctx, float* ms);
/* The part of it spent in cz_execute_frames_kernel (0 when it did not run). */
int cz_context_last_exec_ms(cz_context* ctx, float* ms);
/* With both arenas set, frames the pre-pass prepared completely are executed by cz_execute_frames_kernel (one wave per
 * frame, sequence execution only: no decoders in LDS or registers) — on = 1, the default; on = 0 sends them to
 * cz_decode_frames_kernel's record path instead (the round-2 arrangement, kept for A/B runs: bench.py --no-exec-kernel).
 * Batches that start from a dictionary (cz_context_set_dictionary) always take cz_decode_frames_kernel. */
int cz_context_set_exec_kernel(cz_context* ctx, int on);   /* (on = 4 / 8: that register budget — waves per SIMD — whatever the batch looks like; 1: decided on the device) */
/* With both arenas set, the frames that hold enough sequences to be worth a workgroup can also be executed by cz_wexec_kernel:
 * 16 waves per frame, the output of the block in hand in a 128 KiB LDS window (earlier blocks are read from the output buffer;
 * a block that regenerates more than the window is done in several passes), the three values the reference carries from sequence
 * to sequence (output position, literal cursor, offset history: sequence_execution.cairo:12-129, scratch.cairo:11-19) composed
 * across chunks of 64 sequences by a look-back, match sources ordered byte by byte through a bitmap of final bytes.  It runs SIDE BY
 * SIDE with cz_execute_frames_kernel, on half of the CUs, and the two share the frames (each claims a frame before it starts on
 * it): one is bound by instruction issue, the other by the rate of random reads from HBM.  Whether a batch is executed this way is
 * decided on the device, from the offset-code tables of its blocks (cz_chain_kernel sums them up): only when far offsets outweigh
 * near ones — with near offsets the waves of a frame wait on each other's bytes and a workgroup is no faster than one wave.
 * A frame cz_wexec_kernel cannot finish (a check of execute_sequences fails) goes to cz_decode_frames_kernel, which reports the
 * reference's status.  on = 1, the default; on = 0: cz_execute_frames_kernel alone (A/B runs: bench.py --no-wexec-kernel). */
int cz_context_set_wexec_kernel(cz_context* ctx, int on);
/* A/B knobs: CUs cz_wexec_kernel runs on (0: half of them), frames per such CU that cz_execute_frames_kernel leaves to it at the end
 * of a batch (0: the default, 7), force = 1: side by side whatever the batch's offsets look like (tests use it); force = 2: never
 * cz_wexec_kernel on a near-offset batch (by default it takes such a batch's few large frames — 36 000 sequences and more — when
 * the batch has 2 048 frames or more: each has a CU to itself there instead of a wave among 4 096). */
int cz_context_set_wexec_tuning(cz_context* ctx, int cus, int leave_per_cu, int force);
/* Diagnostics of the most recent batch launch (synchronises): what cz_chain_kernel summed from the blocks' LL / OF / ML code
 * tables, in sequences x 4 — with near offset codes (2..13: offsets below 16 KiB), with far ones, with a literal run above 8 or a
 * match above 16 bytes.  The execute stage is arranged from these on the device: cz_wexec_kernel beside cz_execute_frames_kernel
 * when far > near; the 8-waves-per-SIMD build of cz_execute_frames_kernel when near > far and long < (near + far) / 256. */
int cz_context_last_sequence_stats(cz_context* ctx, uint64_t* near_offsets, uint64_t* far_offsets, uint64_t* long_runs);
/* Diagnostics of the most recent batch launch (synchronises): frames listed for cz_wexec_kernel, frames it finished, frames it
 * handed on to cz_decode_frames_kernel. */
int cz_context_last_wexec_counts(cz_context* ctx, size_t* listed, size_t* finished, size_t* given_up);
/* The chain pre-pass of a batch is two launches of cz_chain_kernel — the LARGE blocks (4 096 sequences and more: a batch lasts as
 * long as its longest chain, sequence_section_decoder.cairo:223-286) on one stream, all others on another — and the execute stage
 * starts behind the second: cz_execute_frames_kernel on every frame without a large block, cz_wexec_kernel on the batch's large
 * frames, block by block behind their chains (frames share nothing: src/frame_decoder.cairo:78-104).  on = 0: one chain launch,
 * the execute stage behind all of it.  Default 0: measured in round 5, the arrangement does not pay yet (profiles/r5/NOTES.md): kept
 * as an experiment, covered by the GPU suite. */
int cz_context_set_early_execute(cz_context* ctx, int on);
/* When the small blocks' chains and every literal of the most recent launch were done, in ms from its start (0: not such a launch). */
int cz_context_last_small_ms(cz_context* ctx, float* ms);
/* Diagnostics of the most recent batch launch (synchronises): entries on the fall-back list — frames the pre-pass and execute
 * kernels handed to cz_decode_frames_kernel, each listed once whoever handed it back.  The list has no analogue in the reference:
 * it is the device-side form of "decode this frame by the reference's own order of steps" (src/frame_decoder.cairo:156-222). */
int cz_context_last_fallback_count(cz_context* ctx, size_t* listed);
/* A batch decode enqueues about 35 stream operations (kernels, events, joins).  When a launch repeats the one before it — the same
 * pointers, sizes and context settings; the bytes behind the pointers may differ — it is captured as a hipGraph, and from then on
 * such a launch is ONE graph submission (what a launch enqueues depends on its arguments and the context alone, never on the data).
 * on = 1 turns that on; default 0: every launch is enqueued operation by operation (measured: 3 % of a Raw/RLE batch, 7 % of a
 * batch of 2 000 frames, nothing on the entropy-coded configurations — profiles/r5/graph_replay_ab.txt).  Nothing in the reference
 * corresponds to this: it is how the loop "decode the next batch into the same buffers" (src/frame_decoder.cairo:156-222 called
 * frame after frame) can be submitted here. */
int cz_context_set_graph_replay(cz_context* ctx, int on);
/* 1 when the most recent batch launch was the replay of a captured graph. */
int cz_context_last_launch_was_replay(const cz_context* ctx);
/* The part of the most recent launch spent in cz_wexec_kernel (0 when it did not run). */
int cz_context_last_wexec_ms(cz_context* ctx, float* ms);

/* Test knobs (0 in normal use).  CZ_DEBUG_CHAIN_CPP_STEP: cz_chain_kernel takes every step of every chain with its plain C++ step
 * instead of the hand-scheduled inline-asm group — the two must leave the same records (sequence_section_decoder.cairo:223-286).
 * CZ_DEBUG_NO_HUF1: cz_huf1_kernel (the one-wave huff0 kernel beside the chain kernel) is not launched, so which kernel decodes a
 * literals section no longer depends on timing (literals_section_decoder.cairo:95-115: cz_huf_kernel hands a frame with an uneven
 * 4-stream split back, cz_huf1_kernel keeps it). */
#define CZ_DEBUG_CHAIN_CPP_STEP 1u
#define CZ_DEBUG_NO_HUF1 2u
#define CZ_DEBUG_WX_POISON 4u       /* cz_wexec_kernel: chunk 2 of every block never publishes its look-back entry, so every wave behind it waits until the
                                       bound of its polling loop (WX_SPIN_LIMIT) and the frame is handed to cz_decode_frames_kernel: the test of that bound */
#define CZ_DEBUG_EXEC_FIRST 8u      /* side by side: cz_execute_frames_kernel is submitted AHEAD of cz_wexec_kernel (normally behind it): the two keep to their halves
                                       of the CUs by the hardware's CU id, so the frames split the same way in either order — the test of that */
int cz_context_set_debug_flags(cz_context* ctx, uint32_t flags);
/* Copies the first `bytes` of the chain arena (headers, state -> code maps and per-sequence records of the most recent batch
 * launch, as cz_chain_kernel left them) to host memory and returns the arena units in use; synchronises.  For tests. */
int cz_context_debug_read_chain_arena(cz_context* ctx, void* dst, size_t bytes, uint64_t* units_in_use);

/* Diagnostic builds only (libcairo_zstd_amd_prof.so, -DCZ_PROFILE): copies out and clears the
 * per-phase shader-cycle sums accumulated by the kernels; returns the number of phases written
 * (0 in the product library, which executes no stamps). */
int cz_context_read_profile(cz_context* ctx, unsigned long long* out, int cap);

/* ------------------------------------------------- stateless header parsers */
/* read_frame_header (src/frame.cairo:152-284) + FrameHeaderTrait::window_size (:106-129). */
typedef struct cz_frame_header {
    uint8_t  descriptor;          /* FrameDescriptor (frame.cairo:25-27) */
    uint8_t  window_descriptor;
    uint8_t  has_dict_id;
    uint8_t  header_len;          /* bytes consumed */
    uint32_t dict_id;
    uint64_t frame_content_size;
    uint64_t window_size;
} cz_frame_header;
/* detail[0..1] = (magic, skip_len) for CZ_E_FH_SKIP_FRAME / CZ_E_FH_BAD_MAGIC; may be NULL. */
int cz_read_frame_header(const uint8_t* src, size_t len, cz_frame_header* out, uint64_t* detail);

/* BlockDecoderTrait::read_block_header (src/decoding/block_decoder.cairo:237-278);
 * BlockHeader (src/blocks/block.cairo:10-15).  Always consumes 3 bytes. */
typedef struct cz_block_header {
    uint8_t  last_block;
    uint8_t  block_type;          /* 0 Raw, 1 RLE, 2 Compressed (3 Reserved is an error) */
    uint32_t decompressed_size;   /* Raw/RLE: size; Compressed: 0 = unknown */
    uint32_t content_size;        /* Raw/Compressed: size; RLE: 1 */
} cz_block_header;
int cz_read_block_header(const uint8_t* src, size_t len, cz_block_header* out);

/* ------------------------------------------------- stream walker */
/* read_frame_header returns a skippable frame to its caller as the error SkipFrame(magic, length)
 * (src/frame.cairo:160-166); cz_stream_split is that caller for a stream of concatenated frames: it cuts the stream
 * into one entry per zstd frame (its end found by walking the block headers, no decoding) and steps over skippable
 * frames, so that a multi-frame .zst goes through cz_decode_batch_* in one launch. */
typedef struct cz_stream_entry {
    uint64_t offset, length;      /* bytes of the stream this entry covers */
    uint32_t kind;                /* CZ_STREAM_FRAME or CZ_STREAM_SKIPPABLE */
    uint32_t magic;               /* skippable frame: its magic number (0x184D2A50..5F) */
    uint64_t content_size;        /* zstd frame: frame_content_size of its header (0 when absent) */
    uint64_t out_bound;           /* zstd frame: Raw / RLE sizes + 128 KiB per compressed block: an output capacity that
                                     holds every frame the zstd format allows (the reference accepts larger blocks: D3) */
    uint64_t window_size;
} cz_stream_entry;
#define CZ_STREAM_FRAME 0u
#define CZ_STREAM_SKIPPABLE 1u
/* Fills entries[0 .. min(*count, cap)); *count = entries in the stream, *consumed = bytes they cover.  Returns CZ_OK
 * when the whole stream was cut up, CZ_E_TARGET_TOO_SMALL when cap < *count, or the status of the header that could
 * not be read (the entries before it are valid). */
int cz_stream_split(const uint8_t* src, size_t len, cz_stream_entry* entries, size_t cap, size_t* count, size_t* consumed);

/* ------------------------------------------------- block level (BlockDecoder + DecoderScratch) */
/* DecoderScratch (src/decoding/scratch.cairo:11-67): the state a frame carries from block to block — Huffman
 * table, three FSE tables + RLE symbols, offset history (1, 4, 8) — and its DecodeBuffer
 * (src/decoding/decode_buffer.cairo:9-15).  Both live in HBM; the handle is the `ref workspace: DecoderScratch` of
 * decode_block_content.  Only the last window_size bytes plus what was not drained yet stay resident. */
typedef struct cz_decoder_scratch cz_decoder_scratch;
int  cz_decoder_scratch_create(cz_context* ctx, uint64_t window_size, cz_decoder_scratch** out);   /* DecoderScratchTrait::new, scratch.cairo:23-40 */
int  cz_decoder_scratch_reset(cz_decoder_scratch* s, uint64_t window_size);                        /* reset, scratch.cairo:42-58 */
void cz_decoder_scratch_destroy(cz_decoder_scratch* s);
size_t cz_decoder_scratch_buffer_len(const cz_decoder_scratch* s);                                 /* buffer.len() */
uint64_t cz_decoder_scratch_total_output(const cz_decoder_scratch* s);                             /* bytes decoded into the buffer so far.  Equals buffer.total_output_counter except
                                                                                                      after matches copied WHOLLY from a dictionary: the reference's counter skips
                                                                                                      those (decode_buffer.cairo:85-90); the device keeps that lag for its window
                                                                                                      test but this getter does not subtract it */
/* DecodeBuffer::drain (decode_buffer.cairo:157-166): moves the whole buffer to dst and feeds the XXH64 state. */
int  cz_decoder_scratch_drain(cz_decoder_scratch* s, uint8_t* dst, size_t cap, size_t* written);
/* DecodeBuffer::drain_to_window_size (:145-155): 1 = Some (bytes beyond window_size moved), 0 = None, < 0 = -cz_status. */
int  cz_decoder_scratch_drain_to_window_size(cz_decoder_scratch* s, uint8_t* dst, size_t cap, size_t* written);
uint64_t cz_decoder_scratch_hash_digest(const cz_decoder_scratch* s);                              /* buffer.hash.digest(): XXH64 of what was drained */

/* ------------------------------------------------- dictionaries */
/* Dictionary (src/decoding/dictionary.cairo:11-18).  The reference parses dictionaries and can seed a DecoderScratch from one,
 * but none of its decoders ever does (frame_decoder.cairo never calls init_from_dict): the same two calls are offered here,
 * on the block level, and a workspace seeded from a dictionary decodes frames that were compressed with it — Repeat-mode /
 * Treeless first blocks use the dictionary's tables, repeat offsets start from its three, matches may reach into its content
 * (decode_buffer.cairo:65-93).  The parsed tables and the content live in HBM. */
typedef struct cz_dictionary cz_dictionary;
/* DictionaryTrait::decode_dict (dictionary.cairo:35-91).  Errors: CZ_E_DICT_BAD_MAGIC (detail[0] = the magic read),
 * CZ_E_DICT_TRUNCATED, or the HuffmanTableError / FSETableError leaf. */
int  cz_dictionary_decode(cz_context* ctx, const uint8_t* raw, size_t len, cz_dictionary** out, uint64_t* detail);
void cz_dictionary_destroy(cz_dictionary* d);
uint32_t cz_dictionary_id(const cz_dictionary* d);                                                 /* Dictionary.id */
size_t cz_dictionary_content_len(const cz_dictionary* d);                                          /* dict_content.len() */
int  cz_dictionary_offset_hist(const cz_dictionary* d, uint32_t out[3]);                           /* Dictionary.offset_hist */
/* DecoderScratchTrait::init_from_dict (scratch.cairo:60-65); call it on a fresh or reset workspace.  reset clears it again
 * (decode_buffer.cairo:38).  The dictionary must stay alive while the workspace uses it. */
int  cz_decoder_scratch_init_from_dict(cz_decoder_scratch* workspace, const cz_dictionary* d);
/* The batch form of the same call: every frame cz_decode_batch_* decodes on this context starts as init_from_dict leaves a
 * workspace (NULL: back to the default, no dictionary).  This is the many-small-records case dictionaries exist for; frames
 * whose first block leans on the dictionary's tables are decoded by cz_decode_frames_kernel alone (the pre-pass lists only
 * frames that define their own tables), the others take the pre-pass as usual.  The dictionary must outlive the launches. */
int  cz_context_set_dictionary(cz_context* ctx, const cz_dictionary* d);

/* BlockDecoder (src/decoding/block_decoder.cairo:20-30): a plain value like the reference's struct. */
typedef struct cz_block_decoder {
    uint8_t internal_state;       /* DecoderState, block_decoder.cairo:26-30 */
    uint8_t header_buffer[3];
} cz_block_decoder;
#define CZ_BLOCK_READY_FOR_HEADER 0   /* ReadyToDecodeNextHeader */
#define CZ_BLOCK_READY_FOR_BODY   1   /* ReadyToDecodeNextBody */
#define CZ_BLOCK_FAILED           2   /* Failed (never set by the reference either, block_decoder.cairo:26-30) */
void cz_block_decoder_new(cz_block_decoder* bd);                                                   /* BlockDecoderTrait::new, :71-75 */
/* read_block_header (:237-278): *consumed = 3 once three bytes were there; moves the state to ReadyToDecodeNextBody. */
int  cz_block_decoder_read_block_header(cz_block_decoder* bd, const uint8_t* src, size_t len, cz_block_header* out, uint8_t* consumed);
/* decode_block_content (:77-137): decodes the body of the block `header` describes from src (positioned just behind the
 * header) into the workspace's DecodeBuffer, on the device.  *consumed = bytes of src the block took (Raw: its size,
 * RLE: 1, Compressed: content_size).  Errors: CZ_E_BLOCK_EXPECTED_HEADER / CZ_E_BLOCK_STATE_FAILED (state machine),
 * CZ_E_BH_RESERVED, or the DecompressBlockError leaf. */
int  cz_block_decoder_decode_block_content(cz_block_decoder* bd, const cz_block_header* header, cz_decoder_scratch* workspace,
                                           const uint8_t* src, size_t len, uint64_t* consumed);

/* ------------------------------------------- 3. resumable single-frame decoder */
/* FrameDecoder / FrameDecoderState (src/frame_decoder.cairo:17-30).  Source and target are
 * HOST buffers (as the reference's ByteArraySlice / ByteArray are); decoding runs on the
 * device, the decoded frame stays resident in HBM until collected. */
typedef struct cz_frame_decoder cz_frame_decoder;

typedef enum cz_strategy {          /* BlockDecodingStrategy, frame_decoder.cairo:33-37 */
    CZ_STRATEGY_ALL = 0,
    CZ_STRATEGY_UPTO_BLOCKS = 1,
    CZ_STRATEGY_UPTO_BYTES = 2
} cz_strategy;

int  cz_frame_decoder_create(cz_context* ctx, cz_frame_decoder** out);
void cz_frame_decoder_destroy(cz_frame_decoder* fd);
/* FrameDecoderStateTrait::new (frame_decoder.cairo:54-76): parses the frame header.
 * *consumed = header bytes; detail as cz_read_frame_header. */
int  cz_frame_decoder_new(cz_frame_decoder* fd, const uint8_t* src, size_t len, size_t* consumed, uint64_t* detail);
/* FrameDecoderStateTrait::reset (:78-104): same, plus the 100 MiB window cap (:92). */
int  cz_frame_decoder_reset(cz_frame_decoder* fd, const uint8_t* src, size_t len, size_t* consumed, uint64_t* detail);
uint64_t cz_frame_decoder_content_size(const cz_frame_decoder* fd);                 /* :125 */
int  cz_frame_decoder_checksum_from_data(const cz_frame_decoder* fd, uint32_t* v);  /* :129; returns 1 = Some */
uint32_t cz_frame_decoder_calculated_checksum(const cz_frame_decoder* fd);          /* :133-138 (low 32 bits of XXH64 of
                                                                                       the bytes drained so far) */
uint64_t cz_frame_decoder_bytes_read_from_source(const cz_frame_decoder* fd);       /* :140 */
int  cz_frame_decoder_is_finished(const cz_frame_decoder* fd);                      /* :144-150 */
size_t cz_frame_decoder_blocks_decoded(const cz_frame_decoder* fd);                 /* :152 */
/* decode_blocks (:156-222).  src points just past what earlier calls consumed.  *consumed =
 * bytes taken by this call, *finished = frame_finished. */
int  cz_frame_decoder_decode_blocks(cz_frame_decoder* fd, const uint8_t* src, size_t len, cz_strategy strategy,
                                    size_t n, size_t* consumed, int* finished);
size_t cz_frame_decoder_can_collect(const cz_frame_decoder* fd);                    /* :233-243 */
/* collect (:224-231).  Returns 1 = Some (bytes moved into dst, *written set), 0 = None,
 * <0 = -cz_status (dst too small: CZ_E_TARGET_TOO_SMALL). */
int  cz_frame_decoder_collect(cz_frame_decoder* fd, uint8_t* dst, size_t cap, size_t* written);
/* read (:328-334); returns bytes moved. */
size_t cz_frame_decoder_read(cz_frame_decoder* fd, uint8_t* dst, size_t cap);
/* FrameDecoderState.decoder_scratch (:25): the frame decoder's own DecoderScratch (owned by the frame decoder). */
cz_decoder_scratch* cz_frame_decoder_scratch(cz_frame_decoder* fd);
/* decode_from_to (:245-326): (*read_len, *written) = (source bytes consumed, target bytes produced). */
int  cz_frame_decoder_decode_from_to(cz_frame_decoder* fd, const uint8_t* src, size_t len, uint8_t* dst, size_t cap,
                                     size_t* read_len, size_t* written);

#ifdef __cplusplus
}
#endif
#endif /* CAIRO_ZSTD_AMD_H */
