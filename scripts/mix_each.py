#!/usr/bin/env python3
"""Diagnostic: the largest frames of the corpus-like mix, each decoded on its own (one wave): execute-kernel time per frame."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import cairo_zstd_amd as cz
from cairo_zstd_amd import synth

n = 12500
top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
b = synth.generate("mix", n, nthreads=16)
order = np.argsort(-b.regen.astype(np.int64))
dev = torch.device("cuda:0")
ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_chain_arena(64 << 20)
ctx.set_literal_arena(32 << 20)
ctx.set_wexec_kernel(os.environ.get("CZ_WEXEC", "1") != "0", force=os.environ.get("CZ_WEXEC", "1") == "force")
torch.cuda.synchronize()                                                # (the context runs on a stream of its own)
t_base = torch.from_numpy(b.base).to(dev)
rows = []
for i in order[:top]:
    idx = np.array([i])
    off, ln, rg = b.off[idx], b.length[idx], b.regen[idx]
    t = [torch.from_numpy(x).to(dev) for x in (off.astype(np.int64), ln.astype(np.int64), np.zeros(1, dtype=np.int64), rg.astype(np.int64))]
    t_out = torch.empty(int(rg[0]) + 256, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ex, wx = [], []
    torch.cuda.synchronize()
    for it in range(3):
        ctx.decode_batch_device(t_base.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), 1, t_out.data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t_res.data_ptr())
        ctx.last_kernel_ms()
        ex.append(ctx.last_exec_ms() + ctx.last_wexec_ms())
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    rows.append((int(i), int(rg[0]), int(ln[0]), int(res["blocks_decoded"][0]), float(np.min(ex)), int(res["status"][0])))
for r in sorted(rows, key=lambda r: -r[4]):
    print(f"frame {r[0]:6d} out {r[1] / 1e6:6.2f} MB in {r[2] / 1e3:7.1f} KB blocks {r[3]:3d} exec {r[4]:7.3f} ms  {r[1] / 1e6 / r[4]:6.2f} GB/s status {r[5]}", flush=True)
ctx.close()
