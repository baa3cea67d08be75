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
