#!/usr/bin/env python3
"""Diagnostic: per-kernel time of one batch decode for a list of library builds.

    python scripts/kernel_times.py <kind> <frames> <lib.so> [<lib.so> ...]

Each library is loaded in a child process (one HIP runtime per process); prints chain-kernel and
total milliseconds (hipEvents inside the library) and whether all frames decoded."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(kind, n):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import cairo_zstd_amd as cz
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from _batches import make_batch
    b, out_off, out_cap, total = make_batch(kind, n)
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(x).to(dev) for x in (b.base, b.off.astype(np.int64), b.length.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
    t_out = torch.empty(total, dtype=torch.uint8, device=dev)
    t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
    if os.environ.get("CZ_PREPASS", "1") == "1":
        ctx.set_chain_arena(int(b.length.sum()) * 6 + (64 << 20))
        if os.environ.get("CZ_LITPASS", "1") == "1":
            ctx.set_literal_arena(int(b.regen.sum()) + (16 << 20))
        cz.lib().cz_context_set_exec_kernel(ctx._h, int(os.environ.get("CZ_EXEC", "1")))   # 0 off, 1 on, 4 / 8: that register budget whatever the batch looks like
        wxe = os.environ.get("CZ_WEXEC", "1").split(",")               # on[,cus[,leave_per_cu[,force]]]
        ctx.set_wexec_kernel(wxe[0] == "1", *(int(v) for v in wxe[1:3]), force=int(wxe[3]) if len(wxe) > 3 else 0)
    if os.environ.get("CZ_GRAPH") is not None and hasattr(ctx, "set_graph_replay"):
        ctx.set_graph_replay(os.environ["CZ_GRAPH"] == "1")
    if os.environ.get("CZ_EARLY") is not None and hasattr(ctx, "set_early_execute"):
        ctx.set_early_execute(os.environ["CZ_EARLY"] == "1")
    tot, ch, ex, lt, wx = [], [], [], [], []
    for it in range(5):
        ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
        tot.append(ctx.last_kernel_ms())
        ch.append(ctx.last_chain_ms())
        ex.append(ctx.last_exec_ms())
        wx.append(ctx.last_wexec_ms())
        lt.append(ctx.last_literals_tail_ms())
    res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
    ok = bool((res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all())
    if os.environ.get("CZ_CHECK", "0") == "1":                          # every frame against the CPU oracle
        import oracle
        got = t_out.cpu().numpy()
        nbad = 0
        for i in range(n):
            st, ref, _ = oracle.decode_frame(b.frame(i), cap=int(out_cap[i]))
            o = int(out_off[i])
            if st != 0 or bytes(got[o:o + len(ref)]) != ref:
                nbad += 1
        ok = ok and nbad == 0
        print(f"oracle check: {nbad} of {n} frames differ", flush=True)
    print(f"{os.path.basename(os.environ.get('CAIRO_ZSTD_AMD_LIB', 'default')):40s} early={os.environ.get('CZ_EARLY', '-')} small done at {ctx.last_small_ms() if hasattr(ctx, 'last_small_ms') else 0:6.3f} ms  total {np.mean(tot[2:]):8.3f} ms  chain {np.mean(ch[2:]):8.3f} ms  "
          f"lit tail {np.mean(lt[2:]):6.3f} ms  wexec {np.mean(wx[2:]):8.3f} ms  exec {np.mean(ex[2:]):8.3f} ms  wexec listed/finished/handed on {ctx.last_wexec_counts()}  near/far/long {ctx.last_sequence_stats()}  execute grid {ctx.execute_grid() if hasattr(cz.lib(), 'cz_context_execute_grid') else '?'}  ok={ok}", flush=True)
    ctx.close()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        kind, n = sys.argv[1], sys.argv[2]
        for lib in sys.argv[3:]:
            env = dict(os.environ, CAIRO_ZSTD_AMD_LIB=os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, n], env=env, timeout=300)
