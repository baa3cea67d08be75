#!/usr/bin/env python3
"""Diagnostic: the corpus-like mix with the chain pre-pass at several chain_min_sequences gates (and without)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import cairo_zstd_amd as cz
from cairo_zstd_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
b = synth.generate("mix", n, nthreads=16)
out_off, out_cap, total = b.out_layout(256)
dev = torch.device("cuda:0")
t = [torch.from_numpy(x).to(dev) for x in (b.base, b.off.astype(np.int64), b.length.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
t_out = torch.empty(total, dtype=torch.uint8, device=dev)
t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
for label, arena, minseq, lit in [("no prepass", 0, 0, 0), ("min 8192", 1, 8192, 0), ("min 2048", 1, 2048, 0), ("min 2048 + literals pass", 1, 2048, 1), ("min 512", 1, 512, 0),
                                  ("min 512 + literals pass", 1, 512, 1), ("min 0", 1, 0, 0)]:
    ctx.set_chain_arena(int(b.length.sum()) * 8 + (64 << 20) if arena else 0, min_sequences=minseq)
    ctx.set_literal_arena(int(b.regen.sum()) + (16 << 20) if lit else 0)
    tot, ch = [], []
    for it in range(4):
        ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
        tot.append(ctx.last_kernel_ms())
        ch.append(ctx.last_chain_ms())
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    ok = bool((res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all())
    print(f"{label:28s} total {np.mean(tot[1:]):8.3f} ms  chain {np.mean(ch[1:]):8.3f} ms  lit tail {ctx.last_literals_tail_ms():6.3f}  ok={ok}", flush=True)
ctx.close()
