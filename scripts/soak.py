#!/usr/bin/env python3
"""Diagnostic: the same batch decoded over and over through the pre-pass pipeline (three streams, seven kernels), every run into a
freshly poisoned buffer and compared on the device with the first run (which is compared with the oracle) — looks for anything
that depends on timing between the kernels.   python scripts/soak.py [repeats=40] [workloads]
CZ_WEXEC=force: every listed frame on cz_wexec_kernel whatever the batch's offset codes say; CZ_WEXEC=0: never.
CZ_GRAPH=1: the repeated launch is captured as a hipGraph and replayed (cz_context_set_graph_replay)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import cairo_zstd_amd as cz
import oracle
from cairo_zstd_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
bad_total = 0
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for kind, n in (("full_4a", 5000), ("mix", 12500), ("huf_literals", 3000), ("raw_rle", 3000), ("full_4b", 1500), ("real", 8000)):
    if only and kind not in only:
        continue
    if kind == "real":                                                  # frames made by the box's libzstd (the bench line's real_libzstd_l3)
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from _batches import make_batch
        b, out_off, out_cap, total = make_batch("real", n)
    else:
        b = synth.generate(kind, n, first_index=777, nthreads=16)
        out_off, out_cap, total = b.out_layout(256)
    t = [torch.from_numpy(x).to(dev) for x in (b.base, b.off.astype(np.int64), b.length.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
    ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_chain_arena(int(b.length.sum()) * 8 + (64 << 20))
    ctx.set_literal_arena(int(b.regen.sum()) + (16 << 20))
    wx = os.environ.get("CZ_WEXEC", "1")
    ctx.set_wexec_kernel(wx != "0", force=wx == "force")
    if os.environ.get("CZ_GRAPH") == "1":
        ctx.set_graph_replay(True)
    ref_out = ref_res = None
    bad = replays = 0
    t_out = torch.empty(total + 256, dtype=torch.uint8, device=dev)      # (the same buffers every run: a launch that repeats can be replayed as a graph)
    t_res = torch.empty(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    for it in range(reps + 1):
        t_out.fill_(0xA5)
        t_res.zero_()
        torch.cuda.synchronize()                                        # torch's default stream has handle 0: the context then runs on a stream of its own, not ordered with the fills above
        ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
        torch.cuda.synchronize()
        replays += int(ctx.last_launch_was_replay())
        if it == 0:
            ref_out, ref_res = t_out.clone(), t_res.clone()
            res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
            o = t_out.cpu().numpy()
            _, olen, ost = oracle.decode_batch(b.base, b.off, b.length, out_off, out_cap, int(total) + 256, nthreads=32)
            ok = bool((res["status"] == np.asarray(ost)).all()) and bool((res["bytes_produced"] == np.asarray(olen)).all())
            ref_o = _
            same = all(o[int(out_off[i]): int(out_off[i]) + int(olen[i])].tobytes() == ref_o[int(out_off[i]): int(out_off[i]) + int(olen[i])].tobytes() for i in range(0, n, 1))
            print(f"{kind:14s} n={n}: first run vs oracle: statuses {ok}, bytes {same}", flush=True)
            bad += 0 if (ok and same) else 1
        elif not (torch.equal(t_out, ref_out) and torch.equal(t_res, ref_res)):
            bad += 1
            d_out = torch.nonzero(t_out != ref_out).flatten()
            d_res = torch.nonzero(t_res != ref_res).flatten()
            msg = f"   run {it} differs from the first: {d_out.numel()} output bytes, {d_res.numel()} result bytes"
            if d_out.numel():
                p0 = int(d_out[0]); f = int(np.searchsorted(out_off, p0, side="right") - 1)
                msg += f"; first output byte {p0} = frame {f} + {p0 - int(out_off[f])} (regen {int(b.regen[f])}), got {int(t_out[p0])} want {int(ref_out[p0])}, last {int(d_out[-1])}"
            if d_res.numel():
                r0 = int(d_res[0]); fr = r0 // cz.RESULT_DTYPE.itemsize
                msg += f"; first result byte {r0} = frame {fr} field offset {r0 % cz.RESULT_DTYPE.itemsize}"
                got = t_res.cpu().numpy().view(cz.RESULT_DTYPE)[fr]; want = ref_res.cpu().numpy().view(cz.RESULT_DTYPE)[fr]
                msg += f"\n      got  {got}\n      want {want}"
            print(msg, flush=True)
    print(f"{kind:14s} {reps} repeats, {bad} differing; cz_wexec_kernel listed / finished / handed on {ctx.last_wexec_counts()}; launches replayed as a graph: {replays}", flush=True)
    bad_total += bad
    ctx.close()
print("TOTAL DIFFERING", bad_total)
sys.exit(1 if bad_total else 0)
