#!/usr/bin/env python3
"""Makes the dictionary fixtures under tests/golden/dict/ with the host's libzstd (data only: a trained dictionary,
frames compressed with it, the originals).  Deterministic inputs; re-running it on another libzstd version may give
other (equally valid) bytes, which is why the outputs are committed.

    dict.bin                 ZDICT_trainFromBuffer over 400 small records (magic 0xEC30A437, tables, content)
    frame_XX.zst / .orig     records compressed with ZSTD_compress_usingDict (level 3 / 19): their first sequences
                             reach into the dictionary content and their first block uses the dictionary's tables
    dict_hist.bin            dict.bin with its three repeat offsets (dictionary.cairo:81-85) patched from (1, 4, 8) — which is
                             also the default history of a reset workspace — to (21, 7, 96)
    hist_XX.zst / .orig      inputs that begin with the dictionary content found 21 / 7 / 96 bytes before its end, compressed with
                             dict_hist.bin: their first sequences are repeat-offset codes, so they decode to .orig only when the
                             workspace really starts from the DICTIONARY's history (checked below: with dict.bin they do not)
"""
import ctypes as C
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dict")
L = C.CDLL("libzstd.so.1")
for f in ("ZDICT_trainFromBuffer", "ZSTD_compress_usingDict", "ZSTD_decompress_usingDict", "ZSTD_compressBound"):
    getattr(L, f).restype = C.c_size_t
L.ZSTD_createCCtx.restype = C.c_void_p
L.ZSTD_createDCtx.restype = C.c_void_p
L.ZSTD_isError.restype = C.c_uint
L.ZDICT_isError.restype = C.c_uint


def record(rng, i):
    """A small JSON-like record: shared field names and vocabulary (what a dictionary is for) + unique values."""
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet", "kilo", "lima"]
    tags = ",".join(f'"{words[int(k)]}"' for k in rng.integers(0, len(words), size=int(rng.integers(2, 7))))
    body = " ".join(words[int(k)] for k in rng.integers(0, len(words), size=int(rng.integers(20, 120))))
    return (f'{{"id": {i}, "user": "user_{int(rng.integers(0, 5000))}", "status": "{"active" if i % 3 else "suspended"}", '
            f'"score": {float(rng.random()):.6f}, "tags": [{tags}], "description": "{body}", '
            f'"checksum": "{int(rng.integers(0, 2**63)):016x}"}}\n').encode()


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261004)
    samples = [record(rng, i) for i in range(400)]
    blob = b"".join(samples)
    sizes = (C.c_size_t * len(samples))(*[len(s) for s in samples])
    dcap = 8192
    dbuf = C.create_string_buffer(dcap)
    n = L.ZDICT_trainFromBuffer(dbuf, C.c_size_t(dcap), blob, sizes, C.c_uint(len(samples)))
    assert not L.ZDICT_isError(C.c_size_t(n)), "ZDICT_trainFromBuffer failed"
    d = dbuf.raw[:n]
    open(os.path.join(OUT, "dict.bin"), "wb").write(d)
    cctx, dctx = C.c_void_p(L.ZSTD_createCCtx()), C.c_void_p(L.ZSTD_createDCtx())
    cases = [(record(rng, 1000 + k), 3) for k in range(4)]
    cases += [(b"".join(record(rng, 2000 + 10 * k + j) for j in range(40)), 19 if k & 1 else 3) for k in range(3)]   # several KB: later matches stay inside the frame
    cases += [(b"".join(record(rng, 3000 + j) for j in range(260)), 3)]                                              # > 128 KiB: two blocks, the second repeats tables
    for k, (orig, level) in enumerate(cases):
        cap = L.ZSTD_compressBound(C.c_size_t(len(orig)))
        cbuf = C.create_string_buffer(cap)
        m = L.ZSTD_compress_usingDict(cctx, cbuf, C.c_size_t(cap), orig, C.c_size_t(len(orig)), d, C.c_size_t(len(d)), C.c_int(level))
        assert not L.ZSTD_isError(C.c_size_t(m))
        z = cbuf.raw[:m]
        back = C.create_string_buffer(len(orig))
        r = L.ZSTD_decompress_usingDict(dctx, back, C.c_size_t(len(orig)), z, C.c_size_t(len(z)), d, C.c_size_t(len(d)))
        assert r == len(orig) and back.raw == orig
        open(os.path.join(OUT, f"frame_{k:02d}.zst"), "wb").write(z)
        open(os.path.join(OUT, f"frame_{k:02d}.orig"), "wb").write(orig)
        print(f"frame_{k:02d}: {len(orig)} -> {len(z)} bytes (level {level})")
    print(f"dict.bin: {len(d)} bytes")
    # ---- the same dictionary with a repeat-offset history that is not the reset default
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    co = oracle.Dictionary(d).info["content_off"]
    hist = (21, 7, 96)
    dh = d[:co - 12] + b"".join(int(h).to_bytes(4, "little") for h in hist) + d[co:]
    open(os.path.join(OUT, "dict_hist.bin"), "wb").write(dh)
    content = d[co:]
    made = 0
    for k, h in enumerate(hist + (21,)):
        tail = record(rng, 5000 + k)
        orig = (content[len(content) - h: len(content) - h + 24] if h >= 24 else (content[len(content) - h:] * 8)[:24]) + tail
        for level in (3, 19):
            cap = L.ZSTD_compressBound(C.c_size_t(len(orig)))
            cbuf = C.create_string_buffer(cap)
            m = L.ZSTD_compress_usingDict(cctx, cbuf, C.c_size_t(cap), orig, C.c_size_t(len(orig)), dh, C.c_size_t(len(dh)), C.c_int(level))
            assert not L.ZSTD_isError(C.c_size_t(m))
            z = cbuf.raw[:m]
            back = C.create_string_buffer(len(orig))
            r = L.ZSTD_decompress_usingDict(dctx, back, C.c_size_t(len(orig)), z, C.c_size_t(len(z)), dh, C.c_size_t(len(dh)))
            assert r == len(orig) and back.raw == orig
            st_h, out_h = oracle.decode_frame_with_dict(z, oracle.Dictionary(dh), cap=len(orig) + 64)
            st_d, out_d = oracle.decode_frame_with_dict(z, oracle.Dictionary(d), cap=len(orig) + 64)
            assert st_h == 0 and out_h == orig
            if st_d == 0 and out_d == orig:
                continue                                          # this frame does not depend on the history: not a fixture
            open(os.path.join(OUT, f"hist_{made:02d}.zst"), "wb").write(z)
            open(os.path.join(OUT, f"hist_{made:02d}.orig"), "wb").write(orig)
            print(f"hist_{made:02d}: offset {h}, level {level}: {len(orig)} -> {len(z)} bytes; with the default history the oracle gives status {st_d}" + ("" if st_d else " and other bytes"))
            made += 1
            break
    assert made >= 2, "no frame depends on the patched history"


if __name__ == "__main__":
    main()
