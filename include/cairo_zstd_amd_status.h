/*
 * cairo_zstd_amd_status.h — status codes shared by the C-ABI (cairo_zstd_amd.h),
 * the HIP kernels (per-frame status words) and the CPU oracle.
 *
 * The reference (NethermindEth/cairo_zstd) reports failures as nested Cairo
 * `Result<_, enum>` values.  A C ABI cannot carry nested enums, so every LEAF
 * variant that the decode path can produce is flattened into one int32 code.
 * Each code cites the reference variant it stands for (file:line relative to
 * the reference tree).  Codes >= 900 have no reference analogue (they belong
 * to the batch / device boundary that the reference does not have).
 *
 * A handful of malformed inputs make the reference *panic* instead of
 * returning Err; those get codes of their own, marked (panic).
 */
#ifndef CAIRO_ZSTD_AMD_STATUS_H
#define CAIRO_ZSTD_AMD_STATUS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cz_status {
    CZ_OK = 0,

    /* ReadFrameHeaderError — src/frame.cairo:141-150 */
    CZ_E_FH_MAGIC_READ = 1,        /* MagicNumberReadError            frame.cairo:157 */
    CZ_E_FH_DESCRIPTOR_READ = 2,   /* FrameDescriptorReadError        frame.cairo:163,174 */
    CZ_E_FH_DICT_ID_READ = 3,      /* DictionaryIdReadError           frame.cairo:207-225; the
                                      reference ALSO returns this variant for a truncated
                                      frame-content-size field (frame.cairo:245-270) */
    CZ_E_FH_WINDOW_DESC_READ = 4,  /* WindowDescriptorReadError       frame.cairo:190 */
    CZ_E_FH_BAD_MAGIC = 5,         /* BadMagicNumber(magic)           frame.cairo:169 */
    CZ_E_FH_SKIP_FRAME = 6,        /* SkipFrame(magic, len)           frame.cairo:165 */

    /* FrameHeaderError — src/frame.cairo:94-102 */
    CZ_E_WINDOW_TOO_BIG = 10,      /* WindowTooBig                    frame.cairo:123 */
    CZ_E_WINDOW_TOO_SMALL = 11,    /* WindowTooSmall                  frame.cairo:126 */

    /* FrameDecoderError — src/frame_decoder.cairo:40-48 */
    CZ_E_WINDOW_SIZE_TOO_BIG = 12, /* WindowSizeTooBig (reset only)   frame_decoder.cairo:93 */
    CZ_E_TARGET_TOO_SMALL = 13,    /* TargetTooSmall (declared, never produced) */

    /* BlockHeaderReadError — src/decoding/block_decoder.cairo:40-58 */
    CZ_E_BH_RESERVED = 20,         /* FoundReservedBlock              block_decoder.cairo:249 */
    CZ_E_BH_SIZE_TOO_LARGE = 21,   /* BlockSizeError::BlockSizeTooLarge block_decoder.cairo:309 */
    CZ_E_BH_TRUNCATED = 22,        /* (panic) r.slice(0,3) on <3 bytes block_decoder.cairo:240 */

    /* DecodeBlockContentError — block_decoder.cairo:60-65 */
    CZ_E_BLOCK_STATE_FAILED = 24,    /* DecoderStateIsFailed          block_decoder.cairo:91 */
    CZ_E_BLOCK_EXPECTED_HEADER = 25, /* ExpectedHeaderOfPreviousBlock block_decoder.cairo:87 */
    CZ_E_BLOCK_TRUNCATED = 26,     /* (panic) body slice out of range block_decoder.cairo:98,105,145 */
    CZ_E_CHECKSUM_TRUNCATED = 27,  /* (panic) source.slice(0,4)       frame_decoder.cairo:192 */

    /* DecompressBlockError — block_decoder.cairo:33-45 */
    CZ_E_MALFORMED_SECTION_HEADER = 30, /* MalformedSectionHeader     block_decoder.cairo:175 */

    /* LiteralsSectionParseError — src/blocks/literals_section.cairo */
    CZ_E_LS_GETBITS = 31,          /* GetBitsError (empty content)    literals_section.cairo:88 */
    CZ_E_LS_NOT_ENOUGH_BYTES = 32, /* NotEnoughBytes                  literals_section.cairo:101 */

    /* DecompressLiteralsError — src/decoding/literals_section_decoder.cairo:18-30 */
    CZ_E_LIT_UNINIT_HUF_TABLE = 40,   /* UninitializedHuffmanTable    :84 */
    CZ_E_LIT_MISSING_JUMP_HEADER = 41,/* MissingBytesForJumpHeader    :93 */
    CZ_E_LIT_MISSING_BYTES = 42,      /* MissingBytesForLiterals      :103 */
    CZ_E_LIT_EXTRA_PADDING = 43,      /* ExtraPadding                 :141,:206 */
    CZ_E_LIT_BITSTREAM_MISMATCH = 44, /* BitstreamReadMismatch        :235 */
    CZ_E_LIT_COUNT_MISMATCH = 45,     /* DecodedLiteralCountMismatch  :174 */

    /* HuffmanTableError — src/huff0/huff0_decoder.cairo:28-43 */
    CZ_E_HUF_SOURCE_EMPTY = 50,       /* SourceIsEmpty                :163 */
    CZ_E_HUF_NOT_ENOUGH_BYTES_FOR_WEIGHTS = 51, /* NotEnoughBytesForWeights :173 */
    CZ_E_HUF_FSE_USED_TOO_MANY_BYTES = 52,      /* FSETableUsedTooManyBytes :183 */
    CZ_E_HUF_EXTRA_PADDING = 53,      /* ExtraPadding                 :224 */
    CZ_E_HUF_TOO_MANY_WEIGHTS = 54,   /* TooManyWeights               :272 (and the u8 overflow
                                         panic at :458 for 256/257 weights) */
    CZ_E_HUF_NOT_ENOUGH_BYTES_IN_SOURCE = 55,   /* NotEnoughBytesInSource :291 */
    CZ_E_HUF_WEIGHT_TOO_BIG = 56,     /* WeightBiggerThanMaxNumBits   :336 */
    CZ_E_HUF_MISSING_WEIGHTS = 57,    /* MissingWeights               :352 */
    CZ_E_HUF_LEFTOVER_NOT_POW2 = 58,  /* LeftoverIsNotAPowerOf2       :360 */
    CZ_E_HUF_MAX_BITS_TOO_HIGH = 59,  /* MaxBitsTooHigh               :386 */

    /* FSETableError — src/fse/fse_decoder.cairo:29-35 */
    CZ_E_FSE_ACC_LOG_TOO_BIG = 60,    /* AccLogTooBig                 :272 */
    CZ_E_FSE_GETBITS = 61,            /* GetBitsError (ran off the end) :267,:349 */
    CZ_E_FSE_PROB_MISMATCH = 62,      /* ProbabilityCounterMismatch   :353 */
    CZ_E_FSE_TOO_MANY_SYMBOLS = 63,   /* TooManySymbols               :358 */

    /* SequencesHeaderParseError — src/blocks/sequence_section.cairo:67-69 */
    CZ_E_SH_NOT_ENOUGH_BYTES = 70,    /* NotEnoughBytes               :82,:90,:96,:102; also the
                                         (panic) read of a missing modes byte at :110 */

    /* DecodeSequenceError — src/decoding/sequence_section_decoder.cairo:20-33 */
    CZ_E_SEQ_EXTRA_PADDING = 80,      /* ExtraPadding                 :63 */
    CZ_E_SEQ_UNSUPPORTED_OFFSET = 81, /* UnsupportedOffset            :130,:236 */
    CZ_E_SEQ_TOO_MANY_BITS = 82,      /* GetBitsError::TooManyBits (LL/ML code out of range) :135,:241 */
    CZ_E_SEQ_TABLE_UNINIT = 83,       /* FSEDecoderError::TableIsUninitialized fse_decoder.cairo:82 */
    CZ_E_SEQ_NOT_ENOUGH_BYTES = 84,   /* NotEnoughBytesForNumSequences :180,:282 */
    CZ_E_SEQ_EXTRA_BITS = 85,         /* ExtraBits                    :191,:293 */
    CZ_E_SEQ_MISSING_RLE_BYTE_LL = 86,/* MissingByteForRleLlTable     sequence_section_decoder.cairo (LL RLE arm) */
    CZ_E_SEQ_MISSING_RLE_BYTE_OF = 87,/* MissingByteForRleOfTable */
    CZ_E_SEQ_MISSING_RLE_BYTE_ML = 88,/* MissingByteForRleMlTable */

    /* ExecuteSequencesError — src/decoding/sequence_execution.cairo:6-10,
       DecodeBufferError — src/decoding/decode_buffer.cairo:18-21 */
    CZ_E_EXEC_NOT_ENOUGH_LITERALS = 90, /* NotEnoughBytesForSequence  sequence_execution.cairo:31 */
    CZ_E_EXEC_ZERO_OFFSET = 91,         /* ZeroOffset                 sequence_execution.cairo:48 */
    CZ_E_EXEC_NOT_ENOUGH_DICT = 92,     /* NotEnoughBytesInDictionary decode_buffer.cairo:70 */
    CZ_E_EXEC_OFFSET_TOO_BIG = 93,      /* OffsetTooBig               decode_buffer.cairo:92 */

    /* ---- DictionaryDecodeError (src/decoding/dictionary.cairo:20-25); its FSETableError / HuffmanTableError
            arms report the leaf codes above ---- */
    CZ_E_DICT_BAD_MAGIC = 95,           /* BadMagicNum                dictionary.cairo:46-48 */
    CZ_E_DICT_TRUNCATED = 96,           /* (panic) word_u32_le(..).expect at :45,:50,:81-83: shorter than its fixed fields */

    /* ---- no reference analogue: batch / device boundary ---- */
    CZ_E_OUTPUT_TOO_SMALL = 900,   /* caller-provided output region cannot hold the frame */
    CZ_E_INVALID_ARG = 901,
    CZ_E_HIP = 902,                /* a HIP runtime call failed; detail = hipError_t */
    CZ_E_UNSUPPORTED = 903,        /* accepted by the reference, refused by the device path
                                      (DESIGN.md "Divergences": Huffman-weight FSE accuracy
                                      log > 9, literals scratch exceeded) */
    CZ_E_NO_DEVICE = 904,          /* library built without / cannot reach a gfx950 device */
    CZ_E_NOT_FINISHED = 905,       /* frame ran out of source before its last block */
    CZ_E_OUT_OF_MEMORY = 906       /* a host allocation failed inside the library (std::bad_alloc does not cross the C ABI) */
} cz_status;

#ifdef __cplusplus
}
#endif
#endif /* CAIRO_ZSTD_AMD_STATUS_H */
