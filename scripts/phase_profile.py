#!/usr/bin/env python3
"""Diagnostic: per-phase shader-cycle shares of cz_decode_frames_kernel (needs the -DCZ_PROFILE
build: make -C cairo_zstd_amd/csrc prof).  Never quote this build's run time; read the SHARES."""
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

PHASES = ["hdr", "huf_build", "huf_decode", "seq_tables", "ring", "chain", "extract", "lit_copy", "match", "raw_rle", "other",
          "(huf_spec)", "(huf_sync)", "(huf_write)", "#chunks_lds", "#chunks_general", "#rounds_general", "#wave_copies"]


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "full_4a"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from _batches import make_batch
    b, out_off, out_cap, total = make_batch(kind, n)
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(x).to(dev) for x in (b.base, b.off.astype(np.int64), b.length.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
    t_out = torch.empty(total, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
    if len(sys.argv) > 3 and sys.argv[3] == "prepass":
        ctx.set_chain_arena(int(b.length.sum()) * 6 + (64 << 20))
        ctx.set_literal_arena(int(b.regen.sum()) + (16 << 20))
    buf = (C.c_uint64 * 64)()
    for it in range(2):
        ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
        torch.cuda.synchronize()
        k = cz.lib().cz_context_read_profile(ctx._h, buf, 64)
    ms = ctx.last_kernel_ms()
    vals = [buf[i] for i in range(k)]
    tot = sum(vals[:11]) or 1
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    print(f"{kind} n={n} kernel {ms:.3f} ms (instrumented), status ok={bool((res['status'] == 0).all())}, launch={ctx.launch_info()}")
    for name, v in zip(PHASES, vals):
        print(f"  {name:11s} {v / n:14.0f} cycles/frame  {100.0 * v / tot:5.1f} %")
    print(f"  total       {tot / n:14.0f} cycles/frame")
    cv = [buf[32 + i] for i in range(8)]
    if sum(cv[:5]):
        ct = sum(cv[:5])
        print(f"cz_chain_kernel wave-time shares (s_memtime ticks summed over waves; {ctx.last_chain_ms():.3f} ms of the launch):")
        for name, v in zip(["parse+tables", "ring fill+init", "top-up events", "chain groups", "finalize"], cv[:5]):
            print(f"  {name:15s} {100.0 * v / ct:5.1f} %")
        print(f"  top-up events {cv[5]}  ({cv[2] / max(cv[5], 1):.0f} ticks each), groups {cv[6]} ({cv[3] / max(cv[6], 1):.1f} ticks each; ticks are shader clocks, ~0.5 ns)")


    rv = [buf[50 + i] for i in range(6)]
    if sum(rv[:4]):
        print(f"  refill events {rv[5]}: pull+stage {rv[0] / max(rv[5], 1):.0f}, own tables {rv[1] / max(rv[5], 1):.0f}, repeat rounds {rv[2] / max(rv[5], 1):.0f}, build {rv[3] / max(rv[5], 1):.0f} ticks each")
    xv = [buf[40 + i] for i in range(10)]
    if sum(xv[:6]):
        xt = sum(xv[:4]) or 1
        print(f"cz_wexec_kernel wave-time shares (s_memtime ticks summed over waves; {ctx.last_wexec_ms():.3f} ms of the launch; {xv[9]} chunks):")
        for name, v in zip(["values+scans", "look-back", "checks+literals", "matches"], xv[:4]):
            print(f"  {name:15s} {100.0 * v / xt:5.1f} %   {v / max(xv[9], 1):10.0f} ticks/chunk")
        print(f"  frame setup+headers {xv[4] / 64 / n:.0f} ticks/frame/wave, flush+tail {xv[5] / 64 / n:.0f} (per wave of 16)")
        print(f"  look-back retries {xv[6]}, match rounds {xv[7]} ({xv[7] / max(xv[9], 1):.2f} per chunk), waits for earlier chunks {xv[8]} ({xv[8] / max(xv[9], 1):.2f} per chunk)")


if __name__ == "__main__":
    main()
