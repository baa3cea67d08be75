#!/usr/bin/env python3
"""Diagnostic: how long the LARGEST frames of the corpus-like mix take on their own (one wave per frame), with / without the pre-pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import cairo_zstd_amd as cz
from cairo_zstd_amd import synth

n = 12500
b = synth.generate("mix", n, nthreads=16)
order = np.argsort(-b.regen.astype(np.int64))
dev = torch.device("cuda:0")
ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
for label, idx in [("largest 1", order[:1]), ("2nd largest", order[1:2]), ("largest 16", order[:16]), ("largest 256", order[:256]), ("largest 3840", order[:3840]),
                   ("all but largest 256", order[256:]), ("all, largest first", order), ("all, index order", np.arange(n))]:
    off, ln, rg = b.off[idx], b.length[idx], b.regen[idx]
    cap = rg.astype(np.uint64)
    pad = (cap + np.uint64(255)) // np.uint64(256) * np.uint64(256)
    ooff = np.zeros(idx.size, dtype=np.uint64)
    ooff[1:] = np.cumsum(pad[:-1])
    total = int(pad.sum())
    t = [torch.from_numpy(x).to(dev) for x in (b.base, off.astype(np.int64), ln.astype(np.int64), ooff.astype(np.int64), cap.astype(np.int64))]
    t_out = torch.empty(total, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(idx.size * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    line = f"{label:22s} frames {idx.size:6d}  MB {rg.sum() / 1e6:8.1f}"
    line += f" max {rg.max() / 1e6:6.2f} MB"
    for pre in (1,):
        ctx.set_chain_arena(int(ln.sum()) * 8 + (64 << 20) if pre else 0)
        ctx.set_literal_arena(int(rg.sum()) + (16 << 20) if pre else 0)
        tot, ch = [], []
        for it in range(3):
            ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), idx.size, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
            tot.append(ctx.last_kernel_ms())
            ch.append(ctx.last_exec_ms() if pre else 0.0)
        res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
        ok = bool((res["status"] == 0).all())
        line += (f" [{ctx.last_prepass_counts(idx.size)[0]} pre-passed]" if pre else "") + f"   {'prepass' if pre else 'single '} total {np.mean(tot[1:]):7.3f} exec {np.mean(ch[1:]):7.3f} ok={ok}"
    print(line, flush=True)
ctx.close()
