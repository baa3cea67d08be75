#!/usr/bin/env python3
"""Diagnostic (not the bench contract): real encoder output instead of the synthetic BASELINE configs — N frames of 128 KiB made by
the box's libzstd (level 3 by default) from text-like and record-like data, decoded in one batch through the pre-pass pipeline;
prints the library's own kernel times and the decoded GB/s, and libzstd's time for the same frames on all cores.
    python scripts/real_data_bench.py [frames=4000] [level=3]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401

import cairo_zstd_amd as cz
import oracle
from conftest import corpus_pairs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = ctypes.CDLL("libzstd.so.1")
L.ZSTD_compress.restype = ctypes.c_size_t
L.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
L.ZSTD_compressBound.restype = ctypes.c_size_t
L.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
rng = np.random.default_rng(7)
text = b"".join(orig for name, z, orig in corpus_pairs(max_orig=20000))
words = [bytes(rng.integers(97, 123, int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(3000)]
SZ = 131072
frames, origs = [], []
cap = L.ZSTD_compressBound(SZ)
dst = ctypes.create_string_buffer(cap)
for i in range(n):
    kind = i % 3
    if kind == 0:                                                       # running text: words by a skewed distribution
        idx = np.minimum(rng.integers(0, len(words), SZ // 4), rng.integers(0, len(words), SZ // 4))
        d = b" ".join(words[int(j)] for j in idx)[:SZ]
    elif kind == 1:                                                     # log-like records
        rec = bytearray()
        t = int(rng.integers(0, 10 ** 9))
        while len(rec) < SZ:
            t += int(rng.integers(1, 50))
            rec += b"%010d host%02d GET /api/v1/item/%06d status=%d bytes=%d\n" % (t, int(rng.integers(0, 40)), int(rng.integers(0, 50000)), (200, 200, 200, 404, 500)[int(rng.integers(0, 5))], int(rng.integers(100, 90000)))
        d = bytes(rec[:SZ])
    else:                                                               # the reference corpus' own originals, rotated
        o = int(rng.integers(0, len(text)))
        d = ((text[o:] + text[:o]) * (SZ // len(text) + 1))[:SZ]
    d = d.ljust(SZ, b".")
    m = L.ZSTD_compress(dst, cap, d, len(d), level)
    frames.append(dst.raw[:m]); origs.append(d)
comp = sum(len(f) for f in frames)
lens = np.array([len(f) for f in frames], dtype=np.uint64)
in_off = np.zeros(n, dtype=np.uint64); in_off[1:] = np.cumsum(lens[:-1])
base = np.frombuffer(b"".join(frames) + b"\0" * 64, dtype=np.uint8)
out_off = np.arange(n, dtype=np.uint64) * np.uint64(SZ)
out_cap = np.full(n, SZ, dtype=np.uint64)
dev = torch.device("cuda:0")
t = [torch.from_numpy(x.copy()).to(dev) for x in (base, in_off.astype(np.int64), lens.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
t_out = torch.empty(n * SZ, dtype=torch.uint8, device=dev)
t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_chain_arena(comp * 8 + (64 << 20))
ctx.set_literal_arena(n * SZ + (16 << 20))
ctx.set_wexec_kernel(os.environ.get("CZ_WEXEC", "1") != "0", force=os.environ.get("CZ_WEXEC", "1") == "force")
torch.cuda.synchronize()                                                # (the context runs on a stream of its own)
ms, ch, ex, wx = [], [], [], []
for it in range(5):
    ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
    ms.append(ctx.last_kernel_ms()); ch.append(ctx.last_chain_ms()); ex.append(ctx.last_exec_ms()); wx.append(ctx.last_wexec_ms())
res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
out = t_out.cpu().numpy()
ok = bool((res["status"] == 0).all()) and all(out[i * SZ:(i + 1) * SZ].tobytes() == origs[i] for i in range(0, n, int(os.environ.get("CZ_CHECK_EVERY", "17"))))
k = float(np.mean(ms[2:]))
print(f"{n} frames of {SZ} B, libzstd level {level}: ratio {n * SZ / comp:.2f}, all decoded and checked: {ok}")
print(f"  GPU step {k:.3f} ms (chain {np.mean(ch[2:]):.3f}, wexec {np.mean(wx[2:]):.3f}, execute {np.mean(ex[2:]):.3f}) = {n * SZ / k / 1e6:.1f} GB/s decoded, {(n * SZ + comp) / k / 1e6:.1f} GB/s algorithmic ({(n * SZ + comp) / k / 1e6 / 8000:.3f} of 8 TB/s)")
t0 = time.time()
good = oracle.libzstd_batch(base, in_off, lens, out_off, out_cap, n * SZ + 64, out_cap, nthreads=os.cpu_count() or 8)
dt = time.time() - t0
print(f"  libzstd on {os.cpu_count()} threads: {good} of {n} frames in {dt * 1e3:.1f} ms = {n * SZ / dt / 1e9:.2f} GB/s")
ctx.close()
