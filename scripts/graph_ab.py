#!/usr/bin/env python3
"""VERDICT r4 item 8 (cheap batches): a repeated batch launch as ONE hipGraph submission against ~35 stream operations.

    python scripts/graph_ab.py [workload frames ...]  > profiles/r5/graph_replay_ab.txt

Per workload, graph replay off / on (cz_context_set_graph_replay): wall-clock ms per step of 30 back-to-back decodes (what bench.py's
ms_per_step measures), the library's own hipEvent figure for a step, whether the launches were replays, outputs compared."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    import torch
    import cairo_zstd_amd as cz
    from _batches import make_batch
    dev = torch.device("cuda:0")
    cases = [("raw_rle", 10000), ("huf_literals", 10000), ("full_4a", 10000), ("mix", 12500)]
    if len(sys.argv) > 2:
        cases = [(sys.argv[i], int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for kind, n in cases:
        b, out_off, out_cap, total = make_batch(kind, n)
        t = [torch.from_numpy(x).to(dev) for x in (b.base, b.off.astype(np.int64), b.length.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
        t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        first = None
        for on in (False, True, False, True):
            ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
            ab, lb = ctx.measure_batch(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t[4].data_ptr())
            ctx.set_chain_arena(ab + (8 << 20))
            ctx.set_literal_arena(lb + (8 << 20))
            ctx.set_graph_replay(on)
            t_out = torch.full((total,), 0xA5, dtype=torch.uint8, device=dev)
            args = (t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
            replays = 0
            for _ in range(4):
                ctx.decode_batch_device(*args)
                replays += ctx.last_launch_was_replay()
            torch.cuda.synchronize()
            steps = 30
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.decode_batch_device(*args)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3 / steps
            ev = ctx.last_kernel_ms()
            parts = (ctx.last_chain_ms(), ctx.last_exec_ms(), ctx.last_wexec_ms())
            res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
            ok = bool((res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all())
            got = t_out.cpu().numpy()
            if first is None:
                first = got
            else:
                ok = ok and bool(np.array_equal(got, first))
            print(f"{kind:14s} {n:6d} frames  graph replay {'on ' if on else 'off'}  wall {wall:7.3f} ms/step  hipEvents {ev:7.3f} ms  chain / execute / cz_wexec_kernel {parts[0]:.3f} / {parts[1]:.3f} / {parts[2]:.3f}"
                  f"  replays among the first 4 launches {replays}  last was replay {ctx.last_launch_was_replay()}  bit-exact {ok}", flush=True)
            ctx.close()
            del t_out
        del t, t_res
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
