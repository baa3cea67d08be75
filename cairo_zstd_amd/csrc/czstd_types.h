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
#define CZ_SCAN_CTL_WORDS 136  /* scan_ctl: [0..31] blocks per size class (class = bit length of nseq), [32..63] fill counters, [64] block work counter,
                                  [72..103] frames per size class (bit length of the compressed size), [104..135] fill counters */

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
    uint32_t* chain_counter; uint32_t chain_min_nseq;
    cz_blk_desc* blk_desc; uint32_t blk_capacity; uint32_t* scan_ctl; uint32_t scan_pass;   /* block list of the pre-pass (cz_scan_kernel) */
    uint32_t* frame_order;                    /* NULL, or the order in which the decode kernels take the frames: largest compressed size first (cz_scan_kernel) */
    uint32_t* exec_counter;                   /* work counter of cz_execute_frames_kernel */
    uint32_t* fallback_list; uint32_t* fallback_count;   /* frames cz_execute_frames_kernel leaves to cz_decode_frames_kernel (NULL: that kernel takes all n frames) */
    /* optional literals pass (cz_decode_frames_kernel with literals_only = 1, launched next to cz_chain_kernel): the
       Huffman-coded literals of every frame the pre-pass takes are decoded into lit_arena — per block a node
       {u64 offset of the next node | 0, u32 regenerated size, u32 0, bytes...}; lit_first[f] = offset of frame f's
       first node, 0 = none: the decode kernels then decode that frame's literals themselves */
    uint8_t* lit_arena; uint64_t lit_capacity; unsigned long long* lit_top; uint64_t* lit_first;
    uint32_t literals_only;
    uint32_t verify_checksum;                 /* batch path: XXH64 of every checksummed frame on the device */
} cz_batch_args;

#endif
