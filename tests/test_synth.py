"""The synthetic frame generator (benchmark inputs) produces frames the oracle decodes to the
declared size; where the system libzstd is available (build container) it must agree too."""
import ctypes
import ctypes.util

import numpy as np
import pytest

import oracle
from cairo_zstd_amd import synth


def _libzstd():
    for name in ("libzstd.so.1", ctypes.util.find_library("zstd")):
        if not name:
            continue
        try:
            z = ctypes.CDLL(name)
            z.ZSTD_decompress.restype = ctypes.c_size_t
            z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
            z.ZSTD_isError.argtypes = [ctypes.c_size_t]
            return z
        except OSError:
            continue
    return None


@pytest.mark.parametrize("kind,n", [("raw_rle", 6), ("huf_literals", 6), ("full_4a", 4), ("full_4b", 2), ("mix", 250)])
def test_generator_roundtrip(kind, n):
    b = synth.generate(kind, n, nthreads=2)
    z = _libzstd()
    rejected = 0
    for i in range(n):
        fr = b.frame(i)
        st, out, info = oracle.decode_frame(fr, cap=int(b.regen[i]) + 16)
        assert st == 0 and len(out) == int(b.regen[i]) and info["consumed"] == len(fr), (kind, i, st)
        if z is not None:
            buf = ctypes.create_string_buffer(int(b.regen[i]) + 64)
            r = z.ZSTD_decompress(buf, len(buf), fr, len(fr))
            if z.ZSTD_isError(r):
                rejected += 1          # e.g. a 2-byte compressed block with no literals: valid for the reference, refused by libzstd
            else:
                assert buf.raw[:r] == out, (kind, i)
    assert rejected <= max(1, n // 100)


def test_generator_is_deterministic_and_thread_independent():
    a = synth.generate("mix", 40, nthreads=1)
    b = synth.generate("mix", 40, nthreads=4)
    assert (a.length == b.length).all() and (a.regen == b.regen).all()
    assert all(a.frame(i) == b.frame(i) for i in range(40))
    c = synth.generate("mix", 20, first_index=20, nthreads=2)
    assert all(c.frame(i) == a.frame(20 + i) for i in range(20))


def test_config_shapes():
    b = synth.generate("full_4a", 2)
    assert (b.regen == 131072).all()
    b = synth.generate("huf_literals", 2)
    assert (b.regen == 131072).all()
    b = synth.generate("raw_rle", 4)
    assert (b.regen == 131072).all() and b.length[1] == 14 and b.length[0] == 10 + 3 + 131072
