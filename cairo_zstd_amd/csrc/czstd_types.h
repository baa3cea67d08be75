/* czstd_types.h — structures shared by the host side (czstd_host.hip) and the kernels. */
#ifndef CZSTD_TYPES_H
#define CZSTD_TYPES_H

#include <stdint.h>
#include "cairo_zstd_amd.h"

#define CZ_FSE_LDS_BYTES ((512 + 512 + 256) * 4)
#ifndef CZ_MAIN_DYN_LDS              /* experiment builds only: a smaller dynamic LDS for batches whose every frame is pre-passed */
#define CZ_MAIN_DYN_LDS CZ_FSE_LDS_BYTES
#endif
#ifndef CZ_MAIN_WAVES                /* waves per SIMD the decode kernel is compiled for (launch bounds) */
#define CZ_MAIN_WAVES 4
#endif
/* dynamic LDS declaration (the CPU emulation harness of tests/emu supplies a static stand-in) */
#ifndef CZ_DYNAMIC_LDS
#define CZ_DYNAMIC_LDS(name) extern __shared__ uint32_t name[]
#endif
#define CZ_WG_THREADS 64                      /* one wavefront per frame */
#ifndef CZ_EXEC_WAVES                /* waves per SIMD cz_execute_frames_kernel is compiled for (launch bounds: caps its registers) */
#define CZ_EXEC_WAVES 4
#endif
#define CZ_EXEC_DYN_LDS (CZ_CHAIN_MAP_BYTES)  /* its dynamic LDS: the state -> code maps of the block in hand */
#define CZ_CHAIN_MAP_BYTES 1280
#define CZ_LIT_SCRATCH_BYTES (256 * 1024 + 256) /* Huffman regenerated size < 2^18 (literals_section.cairo:156-168) */
#define CZ_WG_SCRATCH_BYTES (CZ_LIT_SCRATCH_BYTES + 4096)  /* + the spilled Huffman table of the frame in flight */

/* Carried per-frame decoder state = the reference's DecoderScratch minus the buffers
 * (src/decoding/scratch.cairo:11-19): Huffman table, three FSE tables + RLE symbols,
 * offset history.  Only the resumable frame decoder saves / restores it between launches. */
typedef struct cz_device_frame_state {
    uint16_t huf[2048];
    uint32_t fse[3][512];
    uint32_t hist[3];
    int32_t  fse_rle[3];
    uint8_t  fse_log[3];
    uint8_t  huf_max_bits;
    uint64_t dict_lag;        /* bytes copied by matches that lay wholly in the dictionary: the reference's total_output_counter
                                 does not count them (decode_buffer.cairo:85-90), and its window test uses that counter */
} cz_device_frame_state;

/* One unit of work for the resumable path (cz_frame_decoder_*): decode blocks of ONE frame
 * starting at a block header. */
typedef struct cz_device_task {
    const uint8_t* src; uint64_t src_len;     /* positioned at a block header */
    uint8_t* dst; uint64_t dst_cap;           /* whole-frame output buffer */
    uint64_t produced;                        /* bytes already in dst (total_output_counter) */
    uint64_t drained;                         /* bytes the host already drained (buffer.len = produced - drained) */
    uint64_t window_size;
    uint32_t strategy; uint32_t _pad; uint64_t strategy_n;
    uint32_t has_checksum; uint32_t streaming;/* streaming=1: decode_from_to semantics (stop quietly when short) */
    cz_device_frame_state* state;             /* in/out */
    const uint8_t* dict; uint64_t dict_len;   /* DecodeBuffer.dict_content (decode_buffer.cairo:13): device pointer, 0 = none */
} cz_device_task;

/* One compressed block that has sequences, as cz_scan_kernel lists it for cz_chain_kernel. */
typedef struct cz_blk_desc {
    uint32_t frame;           /* batch entry the block belongs to */
    uint32_t blk_off;         /* offset of the block's content (behind its 3-byte header) in the frame */
    uint32_t bsize;           /* size of the content */
    uint32_t nseq;            /* sequences in the block; 0 = the entry is void (its frame is not pre-passed) */
    uint32_t sbody;           /* offset in the content of the first table description (behind the sequences header) */
    uint32_t modes;           /* the modes byte (sequence_section.cairo:47-57) */
    uint32_t def[3];          /* LL, OF, ML: entry of the earlier block of the frame whose description a Repeat mode refers to; ~0 = not Repeat */
    uint32_t pad;
    uint64_t hdr;             /* index of the block's header in the chain arena */
} cz_blk_desc;
#define CZ_SCAN_CTL_WORDS 224  /* scan_ctl: [0..31] blocks per size class (class = bit length of nseq), [32..63] fill counters, [64] block work counter,
                                  [72..103] frames per size class (bit length of the compressed size), [104..135] fill counters,
                                  [136..167] Huffman literal sections per size class (bit length of the regenerated size), [168..199] fill counters,
                                  [200] work counter of cz_huf_kernel, [201] copy segments counted, [202] placed, [203] work counter of cz_tile_kernel,
                                  [204] frames that are not CZ_PRE_DONE (cz_execute_frames_kernel has nothing to do when there are none),
                                  [205] waves of cz_chain_kernel that have finished (cz_huf1_kernel stops when all have),
                                  [206] frames listed for cz_wexec_kernel (wx_list), [207] of those, frames it did not finish,
                                  [208] listed frames claimed so far (by either execute kernel), [209] frames cz_wexec_kernel finished,
                                  [210] listed frames of CZ_WX_BIG_UNITS and more (the first CZ_WX_BIG_MAX of them carry CZ_PRE_WXBIG),
                                  [211] frames marked CZ_PRE_EARLY, [212] work counter of the small-block launch of cz_chain_kernel (args.chain_part 2),
                                  [213] workgroups of cz_wexec_kernel that stayed (at most args.wx_cus), [214] the same for its early launch */
/* The chain pre-pass runs as TWO launches of cz_chain_kernel (args.chain_part): part 1 the LARGE blocks — CZ_BIG_BLOCK_SEQS sequences and
   more: the head of the block list, which is sorted by size class — part 2 all others.  A batch is as long as its longest chain
   (sequences x ~95 ns), and on ragged batches that is ONE block: with the small blocks in a launch of their own, everything that
   does not depend on a large block is ready at that launch's end, and the execute stage starts on it while the large chains
   still run.  Part 1 publishes every block it finishes (header word 3: 1 done, 2 given up; agent-scope release), so that
   cz_wexec_kernel's early launch — the batch's large frames, a workgroup each — can execute a frame block by block behind its chains. */
#ifndef CZ_BIG_BLOCK_CLASS
#define CZ_BIG_BLOCK_CLASS 13u            /* size class (bit length of the sequence count) from which a block is large */
#endif
#define CZ_BIG_BLOCK_SEQS (1u << (CZ_BIG_BLOCK_CLASS - 1u))   /* 4 096 */
/* cz_wexec_kernel (czstd_wexec.hip): a workgroup per frame, the frame's window in LDS */
#define CZ_WX_RING_LOG 17u
#define CZ_WX_RING (1u << CZ_WX_RING_LOG)   /* frames of at most this many decoded bytes (out_cap) */
#ifndef CZ_WX_BIG_UNITS
#define CZ_WX_BIG_UNITS 36000u            /* ... and from this many on a frame is one of the batch's LARGE frames: on near-offset batches those alone go to cz_wexec_kernel */
#endif
#ifndef CZ_WX_BIG_MIN_FRAMES
#define CZ_WX_BIG_MIN_FRAMES 2048u        /* ... in a batch of at least this many frames, of which at most one in CZ_WX_BIG_SHARE is large (cz_wx_big_only) */
#define CZ_WX_BIG_SHARE 16u
#endif
#define CZ_WX_BIG_MAX 256u                /* at most this many of them (scan_ctl[210] counts the candidates) */
#ifndef CZ_WX_MIN_UNITS
#define CZ_WX_MIN_UNITS 512u                /* chain-arena units (~ sequences) a frame must have to be worth a workgroup */
#endif
/* One Huffman-coded literals section, as cz_scan_kernel lists it for cz_huf_kernel. */
typedef struct cz_lit_seg {
    uint32_t frame;           /* batch entry */
    uint32_t blk_off;         /* offset of the block's content in the frame */
    uint32_t bsize;           /* size of the content */
    uint32_t def;             /* Treeless: list entry of the block whose tree description it reuses; ~0: its own */
    uint64_t dst;             /* where the literals go: offset into the output arena (direct) or into the literal arena (node payload) */
    uint32_t regen;           /* regenerated size */
    uint32_t direct;          /* 1: the block has no sequences and its place in the frame's output is known: the literals ARE its output */
} cz_lit_seg;
/* One run of output bytes whose source and place are known without decoding: a Raw or RLE block, or the Raw / RLE literals of
 * a block without sequences, ahead of the frame's first block with sequences (cz_scan_kernel -> cz_tile_kernel). */
typedef struct cz_copy_seg {
    uint64_t src;             /* offset into the input arena: the bytes (copy) or the one byte (fill) */
    uint64_t dst;             /* offset into the output arena */
    uint32_t len;
    uint32_t fill;            /* 0 copy, 1 fill */
} cz_copy_seg;
#define CZ_PRE_REGULAR 0x80000000u   /* frame_pre[f]: the scan walked the frame to its end and listed all of it; low bits: leading blocks done by the pre-pass kernels */
#define CZ_PRE_DONE    0x40000000u   /* ... and ALL its blocks are done by them: the scan also wrote the frame's result record (no content checksum to verify) */
#define CZ_PRE_EARLY   0x20000000u   /* every block with sequences of the frame is SMALL (below CZ_BIG_BLOCK_SEQS) and the frame is not one of the batch's large ones: its
                                        pre-pass is complete when the small-block chain launch and the literal kernels are (a kernel boundary), long before the
                                        launch that runs the large blocks' chains ends — the early execute launch takes it (cz_execute_frames_kernel, args.early) */
#define CZ_PRE_WXDONE  0x10000000u   /* cz_wexec_kernel finished the frame (result record written): cz_execute_frames_kernel skips it */
#define CZ_PRE_WXLIST  0x08000000u   /* cz_scan_kernel listed the frame for cz_wexec_kernel */
#define CZ_PRE_CLAIMED 0x04000000u   /* cz_wexec_kernel and cz_execute_frames_kernel run side by side and share the frames: whichever sets this bit first does the frame */
#define CZ_PRE_WXBIG   0x02000000u   /* one of the batch's large frames (CZ_WX_BIG_UNITS): on a near-offset batch cz_wexec_kernel does these, and only these */
#define CZ_PRE_LISTED  0x01000000u   /* the frame is on fallback_list (cz_list_fallback: whoever sets this bit first lists it, so a frame is listed ONCE); every kernel of the execute stage skips it */
#define CZ_PRE_COUNT   0x00FFFFFFu

/* chain_top (8 x u64, zeroed per launch): [0] arena units taken; bytes 16.. the work counters of the kernels; [5] / [6] sequences (units of 64, x 256) with
   near / far offset codes, [7] with a literal run above 8 or a match above 16 bytes, summed by cz_chain_kernel (see cz_wx_side_by_side, cz_exec_variant) */
typedef struct cz_batch_args {
    const uint8_t* in_base; const uint64_t* in_off; const uint64_t* in_len;
    uint8_t* out_base; const uint64_t* out_off; const uint64_t* out_cap;
    cz_frame_result* results;
    const cz_device_task* tasks;              /* non-NULL: resumable path, one task per entry */
    uint32_t n;
    uint32_t* work_counter;                   /* zeroed before every launch */
    uint8_t* lit_scratch; uint64_t lit_scratch_stride;   /* one region per resident workgroup */
    const cz_device_frame_state* dict_state; const uint8_t* dict; uint64_t dict_len;   /* batch frames start from this dictionary (cz_context_set_dictionary) */
    unsigned long long* prof;                 /* diagnostic build only: per-phase cycle sums (NULL otherwise) */
    /* optional FSE-chain pre-pass (cz_chain_kernel): NULL / 0 = disabled.  arena[] is in 8-byte units:
       per block {status|nseq<<32, bitstream_off, next header index, 0} then nseq records
       (bit position | LL,ML,OF codes << 32); frame_first[f] = index of frame f's first header, 0 = the
       frame has no chain info and cz_decode_frames_kernel runs the chains itself */
    uint64_t* chain_arena; uint64_t chain_capacity; unsigned long long* chain_top; uint64_t* frame_first;
    uint32_t* chain_counter; uint32_t chain_min_nseq; uint32_t chain_grid;   /* chain_grid: workgroups of the cz_chain_kernel launch */
    cz_blk_desc* blk_desc; uint32_t blk_capacity; uint32_t* scan_ctl; uint32_t scan_pass;   /* block list of the pre-pass (cz_scan_kernel) */
    uint32_t* scan_wave;                      /* 72 words per wave of cz_scan_kernel: what it counted per class in pass 0 */
    uint32_t* frame_order;                    /* NULL, or the order in which the decode kernels take the frames: largest compressed size first (cz_scan_kernel) */
    uint32_t* exec_counter;                   /* work counter of cz_execute_frames_kernel */
    uint32_t* fallback_list; uint32_t* fallback_count;   /* frames cz_execute_frames_kernel leaves to cz_decode_frames_kernel (NULL: that kernel takes all n frames) */
    /* literals and copies of the pre-pass (cz_huf_kernel, cz_tile_kernel next to cz_chain_kernel): cz_scan_kernel lists every
       Huffman-coded literals section and every run of bytes whose place is known without decoding.  Literals of a block that
       has sequences (or whose place is not known) go to lit_arena — per block a node {u64 offset of the next node | 0,
       u32 regenerated size, u32 0, bytes...} laid out and linked by the scan; lit_first[f] = offset of frame f's first node,
       1 = the frame's literals are done and it has no node, 0 = the decode kernels decode that frame's literals themselves.
       frame_pre[f]: CZ_PRE_REGULAR | leading blocks whose output the pre-pass kernels produce; 0 = nothing (the scan or
       cz_huf_kernel met something irregular: cz_decode_frames_kernel does the frame from scratch). */
    uint8_t* lit_arena; uint64_t lit_capacity; unsigned long long* lit_top; uint64_t* lit_first;
    cz_lit_seg* lit_segs; uint32_t lit_seg_capacity; cz_copy_seg* copy_segs; uint32_t copy_seg_capacity; uint32_t* frame_pre;
    uint32_t verify_checksum;                 /* batch path: XXH64 of every checksummed frame on the device */
    uint32_t* wx_list; uint32_t* wx_counter;  /* frames cz_scan_kernel lists for cz_wexec_kernel (NULL: none), and that kernel's work counter */
    uint32_t debug_flags;                     /* CZ_DEBUG_* (cz_context_set_debug_flags): test knobs, 0 in normal use */
    uint32_t exec_variant_force;              /* 0: cz_exec_variant decides between the two register budgets of cz_execute_frames_kernel; 4 / 8: that one */
    uint32_t wx_force;                        /* 0: cz_wx_side_by_side decides from the batch's offset codes; 1: on; 2: off (A/B runs) */
    uint32_t chain_part;                      /* cz_chain_kernel: 0 every listed block; 1 the large ones (and it publishes each block it finishes); 2 the others */
    uint32_t early;                           /* execute kernels: 1 = the EARLY launch, beside the large blocks' chains (cz_execute_frames_kernel: the CZ_PRE_EARLY frames;
                                                 cz_wexec_kernel: the batch's large frames, each block behind its chain's flag); 2 = a later launch of a batch that had early ones
                                                 (every frame is claimed with an atomic before it is executed) */
    uint32_t wx_leave;                        /* cz_execute_frames_kernel leaves the last wx_leave listed frames to cz_wexec_kernel (a frame takes one wave of the former far longer than a workgroup of the latter) */
    uint32_t wx_cus;                          /* workgroups of cz_wexec_kernel that stay (a launch may have more: any beyond this many leave at once); cz_execute_frames_kernel's waves
                                                 on even CUs wait for that many to be in place (cz_cu_side).  (Kept at the END of the struct: the offsets of the fields above
                                                 decide how the compiler spills the execute kernels' scalar registers, and 0.1 ms of config 4a with them: profiles/r5/NOTES.md) */
} cz_batch_args;

#endif
