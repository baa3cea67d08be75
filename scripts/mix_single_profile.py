#!/usr/bin/env python3
"""Diagnostic: phase shares of ONE frame of the corpus-like mix decoded alone (one wave), CZ_PROFILE build.
usage: mix_single_profile.py [rank of the frame by decoded size, default 0 = largest] [lit]"""
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
n = 12500
b = synth.generate("mix", n, nthreads=16)
order = np.argsort(-b.regen.astype(np.int64))
dev = torch.device("cuda:0")
ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
picks = [("rank", int(a)) for a in sys.argv[1:] if a.isdigit()] + [("frame", int(a[1:])) for a in sys.argv[1:] if a[0] == "f" and a[1:].isdigit()]
for how, rank in picks or [("rank", 0)]:
    idx = order[rank:rank + 1] if how == "rank" else np.array([rank])
    off, ln, rg = b.off[idx], b.length[idx], b.regen[idx]
    t = [torch.from_numpy(x).to(dev) for x in (b.base, off.astype(np.int64), ln.astype(np.int64), np.zeros(1, dtype=np.int64), rg.astype(np.int64))]
    t_out = torch.empty(int(rg[0]) + 256, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ctx.set_chain_arena(int(ln.sum()) * 8 + (64 << 20), min_sequences=0)
    ctx.set_literal_arena(int(rg.sum()) + (16 << 20) if "lit" in sys.argv else 0)
    buf = (C.c_uint64 * 64)()
    for it in range(2):
        ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), 1, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
        torch.cuda.synchronize()
        k = cz.lib().cz_context_read_profile(ctx._h, buf, 64)
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    vals = [buf[i] for i in range(11)]
    tot = sum(vals) or 1
    print(f"{how} {rank}: {int(rg[0])} B decoded, {int(ln[0])} B compressed, {int(res['blocks_decoded'][0])} blocks, total {ctx.last_kernel_ms():.3f} ms chain {ctx.last_chain_ms():.3f} ms exec {ctx.last_exec_ms():.3f} ms status {int(res['status'][0])}")
    print("   " + "  ".join(f"{nm} {100.0 * v / tot:.1f}%" for nm, v in zip(PHASES, vals) if v))
    lv = [buf[20 + i] for i in range(11)]
    if sum(lv):
        print("   literals pass: " + "  ".join(f"{nm} {100.0 * v / sum(lv):.1f}%" for nm, v in zip(PHASES, lv) if v) + f"  ({sum(lv) / 2.1e6:.2f} ms of wave time at 2.1 GHz)")
    print(f"   chunks: {buf[14]} on the LDS path, {buf[15]} general with {buf[16]} dependency rounds and {buf[17]} wave-wide match copies")
ctx.close()
