"""Batches for the diagnostics: a synthetic kind of cairo_zstd_amd.synth, or "real" = the real_libzstd_l3 entry of the bench line
(frames made by the box's libzstd at level 3).  Returns (batch, out_off, out_cap, total)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_batch(kind, n, pad=256):
    if kind != "real":
        from cairo_zstd_amd import synth
        b = synth.generate(kind, n, nthreads=16)
        return (b,) + tuple(b.out_layout(pad))
    sys.path.insert(0, ROOT)
    import bench
    frames, origs, _ = bench._real_frames(n)
    length = np.array([len(f) for f in frames], dtype=np.uint64)
    b = types.SimpleNamespace(base=np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy(), length=length, n=n,
                              off=np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64), regen=np.array([len(o) for o in origs], dtype=np.uint64))
    b.frame = lambda i: bytes(b.base[int(b.off[i]): int(b.off[i]) + int(b.length[i])])   # (as the arrays are now: big_parity.py damages them)
    out_cap = (b.regen + pad).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(out_cap)[:-1]]).astype(np.uint64)
    return b, out_off, out_cap, int(out_cap.sum())
