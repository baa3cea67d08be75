#!/usr/bin/env python3
"""Diagnostic (needs the -DCZ_PROFILE build): phase shares of cz_decode_frames_kernel on the K largest frames of the mix."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CAIRO_ZSTD_AMD_LIB"] = os.path.join(ROOT, "cairo_zstd_amd", "csrc", "libcairo_zstd_amd_prof.so")
import numpy as np
import torch

import cairo_zstd_amd as cz
from cairo_zstd_amd import synth

PHASES = ["hdr", "huf_build", "huf_decode", "seq_tables", "ring", "chain", "extract", "lit_copy", "match", "raw_rle", "other"]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
b = synth.generate("mix", 12500, nthreads=16)
idx = np.argsort(-b.regen.astype(np.int64))[:K]
off, ln, rg = b.off[idx], b.length[idx], b.regen[idx]
cap = rg.astype(np.uint64)
pad = (cap + np.uint64(255)) // np.uint64(256) * np.uint64(256)
ooff = np.zeros(idx.size, dtype=np.uint64)
ooff[1:] = np.cumsum(pad[:-1])
dev = torch.device("cuda:0")
t = [torch.from_numpy(x).to(dev) for x in (b.base, off.astype(np.int64), ln.astype(np.int64), ooff.astype(np.int64), cap.astype(np.int64))]
t_out = torch.empty(int(pad.sum()), dtype=torch.uint8, device=dev)
t_res = torch.zeros(K * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
buf = (C.c_uint64 * 64)()
for it in range(2):
    ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), K, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
    torch.cuda.synchronize()
    k = cz.lib().cz_context_read_profile(ctx._h, buf, 64)
vals = [buf[i] for i in range(11)]
tot = sum(vals) or 1
print(f"largest {K} frames of mix, kernel {ctx.last_kernel_ms():.3f} ms (instrumented)")
for name, v in zip(PHASES, vals):
    print(f"  {name:11s} {100.0 * v / tot:5.1f} %   {v / K:12.0f} cycles/frame")
ctx.close()
