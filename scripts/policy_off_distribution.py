#!/usr/bin/env python3
"""the execute stage is arranged on the device from thresholds that were fitted on the bench's own batches
(cz_wx_side_by_side: far > near; cz_wx_big_only: >= 2 048 frames, at most one in sixteen large, 36 000 sequences; up to 896 listed
frames all on cz_wexec_kernel).  This runs batches the thresholds were NOT fitted on and compares what the device decides (auto)
with both forced settings:

    python scripts/policy_off_distribution.py  > profiles/r5/policy_off_distribution.txt

Batches: config 4a at 1 500 / 3 000 / 20 000 frames, the corpus-like mix at 2 000 / 25 000 frames and with its large frames doubled,
real frames made by the box's libzstd at level 19.  Settings: auto; cz_wexec_kernel off (cz_execute_frames_kernel alone); side by
side forced.  Per batch: ms per decode (hipEvents in the library, mean of 3 after 2 warm-ups), the frames cz_wexec_kernel finished,
and auto's distance from the better forced setting.  Every output is compared with the first setting's (and that one with the
oracle on a sample)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def batch_from_frames(frames, regen, pad=256):
    length = np.array([len(f) for f in frames], dtype=np.uint64)
    b = types.SimpleNamespace(base=np.frombuffer(b"".join(frames) + bytes(64), dtype=np.uint8).copy(), length=length, n=len(frames),
                              off=np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.uint64), regen=np.asarray(regen, dtype=np.uint64))
    out_cap = (b.regen + pad).astype(np.uint64)
    out_off = np.concatenate([[0], np.cumsum(out_cap)[:-1]]).astype(np.uint64)
    return b, out_off, out_cap, int(out_cap.sum())


def make(kind, n):
    from _batches import make_batch
    if kind == "mix_large_doubled":
        b, out_off, out_cap, total = make_batch("mix", n)
        big = np.argsort(b.regen)[-max(1, n * 145 // 12500):]               # the share of the bench's mix that is large, once more
        frames = [bytes(b.base[int(b.off[i]): int(b.off[i]) + int(b.length[i])]) for i in range(b.n)] + [bytes(b.base[int(b.off[i]): int(b.off[i]) + int(b.length[i])]) for i in big]
        return batch_from_frames(frames, list(b.regen) + [b.regen[i] for i in big])
    if kind == "real_l19":
        import bench
        frames, origs, _ = bench._real_frames(n, level=19, distinct=192)
        return batch_from_frames(frames, [len(o) for o in origs])
    return make_batch(kind, n)


def main():
    import torch
    import cairo_zstd_amd as cz
    import oracle
    dev = torch.device("cuda:0")
    cases = [("full_4a", 1500), ("full_4a", 3000), ("full_4a", 20000), ("mix", 2000), ("mix", 25000), ("mix_large_doubled", 12500), ("real_l19", 8000)]
    if len(sys.argv) > 1:
        cases = [(sys.argv[i], int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    worst = 0.0
    for kind, n in cases:
        b, out_off, out_cap, total = make(kind, n)
        n = b.n
        t = [torch.from_numpy(x).to(dev) for x in (b.base, b.off.astype(np.int64), b.length.astype(np.int64), out_off.astype(np.int64), out_cap.astype(np.int64))]
        t_res = torch.zeros(n * cz.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        rows, first = [], None
        for name, on, force in (("auto", True, 0), ("cz_wexec_kernel off", False, 0), ("side by side forced", True, 1)):
            ctx = cz.Context(0, torch.cuda.current_stream().cuda_stream)
            ab, lb = ctx.measure_batch(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t[4].data_ptr())
            ctx.set_chain_arena(ab + (8 << 20))
            ctx.set_literal_arena(lb + (8 << 20))
            ctx.set_wexec_kernel(on, force=force)
            t_out = torch.full((total,), 0xA5, dtype=torch.uint8, device=dev)
            ms = []
            for it in range(5):
                ctx.decode_batch_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), n, t_out.data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t_res.data_ptr())
                ms.append(ctx.last_kernel_ms())
            res = t_res.cpu().numpy().view(cz.RESULT_DTYPE)
            ok = bool((res["status"] == 0).all() and (res["bytes_produced"] == b.regen).all())
            got = t_out.cpu().numpy()
            if first is None:
                first = got
                for i in list(range(0, n, max(1, n // 24)))[:24]:           # the first setting against the oracle, a sample
                    st, ref, _ = oracle.decode_frame(bytes(b.base[int(b.off[i]): int(b.off[i]) + int(b.length[i])]), cap=int(out_cap[i]))
                    o = int(out_off[i])
                    ok = ok and st == 0 and bytes(got[o:o + len(ref)]) == ref
            else:
                ok = ok and bool(np.array_equal(got, first))
            rows.append((name, float(np.mean(ms[2:])), ctx.last_wexec_counts(), ctx.last_sequence_stats(), ok))
            ctx.close()
            del t_out
        best_forced = min(r[1] for r in rows[1:])
        gap = rows[0][1] / best_forced - 1.0
        worst = max(worst, gap)
        print(f"== {kind}, {n} frames: near / far / long sums {rows[0][3]}")
        for name, m, wc, _, ok in rows:
            print(f"   {name:22s} {m:8.3f} ms   cz_wexec_kernel listed / finished / handed on {wc}   bit-exact {ok}")
        print(f"   auto is {100.0 * gap:+.1f} % from the better forced setting ({best_forced:.3f} ms)")
        del t, t_res
        torch.cuda.empty_cache()
    print(f"worst case: auto {100.0 * worst:+.1f} % behind the better forced setting")


if __name__ == "__main__":
    main()
