#!/usr/bin/env python3
"""Feasibility study for segment-speculative FSE chains.  CPU only, oracle only (test infrastructure).

A block's sequences section is ONE chain of three interleaved FSE state machines plus raw extra bits
(sequence_section_decoder.cairo:223-286).  cz_chain_kernel could cut a block into segments only if a decoder that starts at a
guessed (bit position, states) falls onto the true trajectory soon.  This script measures that: for every block with >= 1 024
sequences, `starts` decoders begin at random bit positions (states = what an initialisation at that position reads) and run until
they stand on a true sequence boundary with equivalent states (oracle/zstd_oracle.c: czo_fse_sync_study).
    python scripts/fse_sync_study.py > profiles/r4/fse_sync_distance.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle
from cairo_zstd_amd import synth

MAX_STEPS = 8192
L = oracle.lib()
L.czo_fse_sync_study.restype = C.c_int
L.czo_fse_sync_study.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]


def study(name, frames, caps, starts=64):
    hist = np.zeros(MAX_STEPS + 2, dtype=np.uint64)
    summ = np.zeros(8, dtype=np.uint64)
    for i, (fr, cap) in enumerate(zip(frames, caps)):
        b = np.frombuffer(fr, dtype=np.uint8)
        e = L.czo_fse_sync_study(b.ctypes.data, len(fr), cap, 1024, starts, MAX_STEPS, 1234 + i, hist.ctypes.data, summ.ctypes.data)
        assert e == 0, (name, i, e)
    n = int(summ[1])
    if not n:
        print(f"{name}: no block with >= 1024 sequences"); return
    cum = np.cumsum(hist[:MAX_STEPS + 1])
    def q(p):
        k = int(np.searchsorted(cum, p * n))
        return k if k <= MAX_STEPS else None
    conv = int(cum[-1])
    print(f"{name}: {int(summ[0])} blocks, {int(summ[2])} sequences ({int(summ[3]) / max(int(summ[2]), 1):.1f} bits per sequence), {n} speculative starts")
    print(f"   converged within {MAX_STEPS} steps: {conv} ({100.0 * conv / n:.2f} %); within 256: {100.0 * int(cum[256]) / n:.2f} %; within 2048: {100.0 * int(cum[2048]) / n:.2f} %")
    print(f"   steps to convergence: median {q(0.5)}, 90 % {q(0.9)}, 99 % {q(0.99)}   (None = not within {MAX_STEPS})")
    print(f"   steps on a true sequence boundary with different states: {int(summ[4])} ({int(summ[4]) / n:.1f} per start)")


def main():
    b = synth.generate("full_4a", 24, nthreads=4)
    study("config 4a (synthetic: ll = 1, ml = 3, FSE-compressed tables)", [b.frame(i) for i in range(b.n)], [int(r) + 8 for r in b.regen])
    b = synth.generate("mix", 600, nthreads=4)
    study("mix (corpus-like synthetic frames)", [b.frame(i) for i in range(b.n)], [int(r) + 8 for r in b.regen])
    try:
        z = C.CDLL("libzstd.so.1")
    except OSError:
        print("libzstd not present: no real-encoder frames"); return
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    from conftest import corpus_pairs
    rng = np.random.default_rng(7)
    text = b"".join(orig for name, zz, orig in corpus_pairs(max_orig=20000))
    words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(3000)]
    SZ = 131072
    datas = []
    for i in range(36):
        kind = i % 3
        if kind == 0:
            idx = np.minimum(rng.integers(0, len(words), SZ // 4), rng.integers(0, len(words), SZ // 4))
            d = b" ".join(words[int(j)] for j in idx)[:SZ]
        elif kind == 1:
            rec = bytearray(); t = int(rng.integers(0, 10 ** 9))
            while len(rec) < SZ:
                t += int(rng.integers(1, 50))
                rec += b"%010d host%02d GET /api/v1/item/%06d status=%d bytes=%d\n" % (t, int(rng.integers(0, 40)), int(rng.integers(0, 50000)), (200, 200, 200, 404, 500)[int(rng.integers(0, 5))], int(rng.integers(100, 90000)))
            d = bytes(rec[:SZ])
        else:
            o = int(rng.integers(0, len(text)))
            d = ((text[o:] + text[:o]) * (SZ // len(text) + 1))[:SZ]
        datas.append(d.ljust(SZ, b"."))
    for level in (3, 19):
        frames = []
        dst = C.create_string_buffer(SZ * 2)
        for d in datas:
            m = z.ZSTD_compress(dst, SZ * 2, d, len(d), level)
            frames.append(dst.raw[:m])
        study(f"libzstd level {level}, 128 KiB frames of text / log records / corpus originals", frames, [SZ + 8] * len(frames))


if __name__ == "__main__":
    main()
