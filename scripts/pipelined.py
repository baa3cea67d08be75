#!/usr/bin/env python3
"""two batches in flight.  Two contexts, each with arenas, streams and output of its own, decode the same kind of
batch alternately without waiting for each other: call i + 1's pre-pass (cz_chain_kernel: one wave per SIMD, most of the chip's issue
slots idle) overlaps call i's execute stage.  Reports ms per batch in the steady state beside the one-batch-at-a-time figure.

    GPU_MAX_HW_QUEUES=8 python scripts/pipelined.py [workload] [frames] [depth]

(The runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default, and two streams on one queue run one
after the other: two contexts x three streams need the larger number.  The script sets it itself when it is not set.)"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import time

import numpy as np
import torch

import cairo_zstd_amd as cz
from _batches import make_batch


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "full_4a"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    b, out_off, out_cap, total = make_batch(kind, n)
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(x).to(dev) for x in (b.base, b.off.astype(np.int64), b.length.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
    lanes = []
    for k in range(depth):
        s = torch.cuda.Stream()
        ctx = cz.Context(0, s.cuda_stream)
        ab, lb = ctx.measure_batch(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t[4].data_ptr())
        ctx.set_chain_arena(ab + (8 << 20))
        ctx.set_literal_arena(lb + (8 << 20))
        lanes.append((s, ctx, torch.full((total,), 0xA5, dtype=torch.uint8, device=dev), torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)))
    torch.cuda.synchronize()

    def decode(k):
        s, ctx, t_out, t_res = lanes[k]
        ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())

    for k in range(depth):                                               # warm-up, one at a time
        for _ in range(2):
            decode(k)
        torch.cuda.synchronize()
    single = []
    for _ in range(6):
        t0 = time.perf_counter(); decode(0); torch.cuda.synchronize(); single.append((time.perf_counter() - t0) * 1e3)
    one_kernel = lanes[0][1].last_kernel_ms()
    reps = 12
    t0 = time.perf_counter()
    for i in range(reps):
        decode(i % depth)                                                # enqueue only: the contexts' streams run side by side
    torch.cuda.synchronize()
    piped = (time.perf_counter() - t0) * 1e3 / reps
    ok = True
    first = lanes[0][2].cpu().numpy()
    for k in range(depth):
        res = lanes[k][3].cpu().numpy().view(cz.RESULT_DTYPE)
        ok = ok and bool((res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all()) and bool(np.array_equal(lanes[k][2].cpu().numpy(), first))
    print(f"{kind} {n} frames: one batch at a time {np.median(single):.3f} ms wall ({one_kernel:.3f} ms by the library's events); {depth} batches in flight: {piped:.3f} ms per batch "
          f"({float(b.regen.sum()) / (piped * 1e-3) / 1e9:.1f} GB/s decompressed); all outputs identical and complete: {ok}; GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
    import json
    print("JSON " + json.dumps({"workload": kind, "frames": n, "batches_in_flight": depth, "ms_per_batch_one_at_a_time": float(np.median(single)), "ms_per_batch_pipelined": piped,
                                "decompressed_GBps_pipelined": float(b.regen.sum()) / (piped * 1e-3) / 1e9, "outputs_identical_and_complete": ok,
                                "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}))
    for s, ctx, _, _ in lanes:
        ctx.close()


if __name__ == "__main__":
    main()
