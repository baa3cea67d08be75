"""C-ABI checks that need no GPU: the library loads, exports every entry point the header
declares, refuses to create a context without a device (no CPU fallback), and its stateless
header parsers agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np

import cairo_zstd_amd as cz
import oracle
from conftest import ROOT, corpus_pairs


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "cairo_zstd_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cz_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    names = _declared_functions()
    assert len(names) >= 25
    L = cz.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.cz_abi_version() == 1


def test_status_header_matches_python_names():
    assert cz.status.CZ_OK == 0 and cz.status.name(cz.status.CZ_E_SEQ_EXTRA_BITS) == "CZ_E_SEQ_EXTRA_BITS"
    assert len(set(cz.status.CODES.values())) == len(cz.status.CODES)


def test_no_device_means_no_decode():
    import torch
    if torch.cuda.is_available():
        return
    try:
        cz.Context(0)
    except cz.CzError as e:
        assert e.code == cz.status.CZ_E_NO_DEVICE
    else:
        raise AssertionError("a context was created without a GPU")


def test_frame_header_parser_matches_oracle():
    for name, z, orig in corpus_pairs():
        st, fh, _ = cz.read_frame_header(z)
        ost, out, info = oracle.decode_frame(z, cap=len(orig) + 8)
        assert st == 0 and ost == 0
        assert fh.window_size == info["window_size"] and fh.frame_content_size == info["content_size"], name
        for cut in range(0, fh.header_len):
            a = cz.read_frame_header(z[:cut])[0]
            b = oracle.FrameDecoder().new(z[:cut])[0]
            assert a == b and a != 0, (name, cut)
    skip = bytes.fromhex("5a2a4d18") + (1234).to_bytes(4, "little")
    st, _, detail = cz.read_frame_header(skip)
    assert st == cz.status.CZ_E_FH_SKIP_FRAME and detail == (0x184D2A5A, 1234)


def test_block_header_parser():
    # block_decoder.cairo:237-321: last bit, type, 21-bit size; RLE content is 1 byte
    st, bh = cz.read_block_header(bytes([0b101, 0x00, 0x02]))          # last, type 2, size 2^14
    assert st == 0 and (bh.last_block, bh.block_type, bh.content_size, bh.decompressed_size) == (1, 2, 1 << 14, 0)
    st, bh = cz.read_block_header(bytes([0b010 | (5 << 3), 0, 0]))      # RLE, size 5
    assert st == 0 and (bh.block_type, bh.content_size, bh.decompressed_size) == (1, 1, 5)
    assert cz.read_block_header(bytes([0b110, 0, 0]))[0] == cz.status.CZ_E_BH_RESERVED
    assert cz.read_block_header(bytes([0x08, 0x00, 0x10]))[0] == cz.status.CZ_E_BH_SIZE_TOO_LARGE   # 131073
    assert cz.read_block_header(bytes([0x00, 0x00, 0x10]))[0] == 0                                   # 131072 is allowed
    assert cz.read_block_header(b"\x00\x00")[0] == cz.status.CZ_E_BH_TRUNCATED


def test_stream_split_walks_frames_and_skippable_frames():
    """cz_stream_split (host side): the caller's half of SkipFrame (src/frame.cairo:160-166)."""
    pairs = corpus_pairs(max_orig=3000)[:4]
    skip = bytes.fromhex("5a2a4d18") + (5).to_bytes(4, "little") + b"hello"
    parts = [pairs[0][1], skip, pairs[1][1], pairs[2][1], skip, skip, pairs[3][1]]
    data = b"".join(parts)
    st, ents, consumed = cz.stream_split(data)
    assert st == 0 and consumed == len(data) and len(ents) == len(parts)
    pos = 0
    for e, p in zip(ents, parts):
        assert int(e["offset"]) == pos and int(e["length"]) == len(p)
        assert int(e["kind"]) == (cz.STREAM_SKIPPABLE if p is skip else cz.STREAM_FRAME)
        pos += len(p)
    assert all(int(e["magic"]) == 0x184D2A5A for e in ents if int(e["kind"]) == cz.STREAM_SKIPPABLE)
    for e, (name, z, orig) in zip(ents[ents["kind"] == cz.STREAM_FRAME], pairs):
        assert int(e["out_bound"]) >= len(orig)
    # a cut inside the last frame: the entries before it stay valid, the status says what could not be read
    st, ents2, consumed2 = cz.stream_split(data[:-3])
    assert st == cz.status.CZ_E_CHECKSUM_TRUNCATED and len(ents2) == len(parts) - 1 and consumed2 == len(data) - len(parts[-1])
    st, ents3, _ = cz.stream_split(data + b"\x00\x01\x02\x03\x04")
    assert st == cz.status.CZ_E_FH_BAD_MAGIC and len(ents3) == len(parts)
    assert cz.stream_split(b"")[0] == 0
    # a skippable frame that claims more bytes than there are
    assert cz.stream_split(bytes.fromhex("502a4d18") + (9).to_bytes(4, "little") + b"abc")[0] == cz.status.CZ_E_BLOCK_TRUNCATED


def test_block_decoder_state_machine_without_a_device():
    """BlockDecoder's two-state machine (src/decoding/block_decoder.cairo:26-30, :86-93, :271) is host logic."""
    bd = cz.BlockDecoder()
    assert bd.internal_state == cz.BlockDecoder.READY_FOR_HEADER
    st, bh, used = bd.read_block_header(bytes([0b110, 0, 0]))            # Reserved
    assert st == cz.status.CZ_E_BH_RESERVED and used == 3 and bd.internal_state == cz.BlockDecoder.READY_FOR_HEADER
    st, bh, used = bd.read_block_header(b"\x00\x00")
    assert st == cz.status.CZ_E_BH_TRUNCATED and used == 0
    st, bh, used = bd.read_block_header(bytes([0b010 | (5 << 3), 0, 0]))  # RLE, size 5
    assert st == 0 and used == 3 and bd.internal_state == cz.BlockDecoder.READY_FOR_BODY
    assert (bh.block_type, bh.content_size, bh.decompressed_size) == (1, 1, 5)


def test_partition_balanced_matches_the_python_dealing():
    """cz_partition_balanced (the dealing cz_decode_batch_multi uses; no device needed) == cairo_zstd_amd.dist.partition_balanced,
    which bench.py uses across ranks: heaviest first, to the lightest part, ties to the lower index."""
    import numpy as np
    import cairo_zstd_amd as cz
    from cairo_zstd_amd import dist as czdist
    rng = np.random.default_rng(3)
    for n, parts in ((0, 3), (1, 8), (7, 8), (1000, 8), (257, 2), (64, 1)):
        w = np.concatenate([rng.integers(1, 300, n - n // 10), rng.integers(50_000, 900_000, n // 10)]).astype(np.uint64) if n else np.zeros(0, np.uint64)
        got = cz.partition_balanced(w, parts)
        want = czdist.partition_balanced(w, parts)
        assert (got == want).all(), (n, parts)
        if n >= 100:
            loads = np.array([w[got == r].sum() for r in range(parts)], dtype=np.float64)
            assert loads.max() / loads.mean() < 1.05
