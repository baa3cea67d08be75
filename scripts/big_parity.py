#!/usr/bin/env python3
"""One-off large parity sweep on the GPU (diagnostic): synthetic batches through the C ABI with and
without the chain pre-pass (forced on every frame), every frame compared with the oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import cairo_zstd_amd as cz
import oracle
from cairo_zstd_amd import synth


def main():
    total_bad = 0
    sets = (("mix", 6000, 100000), ("mix", 6000, 300000), ("mix!", 6000, 500000), ("full_4a", 300, 5000), ("full_4b", 150, 7000), ("huf_literals", 200, 9000),
            ("raw_rle", 300, 11000), ("real", 1536, 1), ("real!", 1536, 2))
    only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
    for kind, n, first in sets:
        if only and kind not in only:
            continue
        damaged = kind.endswith("!")                                     # every third frame gets one flipped bit or loses its tail
        kind = kind.rstrip("!")
        if kind == "real":                                               # frames made by the box's libzstd (the bench line's real_libzstd_l3)
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            from _batches import make_batch
            b = make_batch("real", n, pad=0)[0]
        else:
            b = synth.generate(kind, n, first_index=first, nthreads=16)
        if damaged:
            rng = np.random.default_rng(first)
            for i in range(0, n, 3):
                o, ln = int(b.off[i]), int(b.length[i])
                if rng.integers(0, 4) == 0:
                    b.length[i] = max(1, ln - int(rng.integers(1, min(ln, 4000))))
                else:
                    b.base[o + int(rng.integers(4, ln))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        frames = [b.frame(i) for i in range(n)]
        caps = [int(r) for r in b.regen]                                 # the capacities the oracle gets below (out_layout)
        if kind == "real":
            o_cap = b.regen.astype(np.uint64)
            o_off = (np.concatenate([[0], np.cumsum((o_cap + 63) // 64 * 64)[:-1]])).astype(np.uint64)
            o_total = int(o_off[-1] + o_cap[-1])
        else:
            o_off, o_cap, o_total = b.out_layout(64)
        _, olen, ost = oracle.decode_batch(b.base, b.off, b.length, o_off, o_cap, int(o_total) + 256, nthreads=32)
        ref_out = _
        for prepass in (0, 1, 2):                                        # 2: pre-pass + literals pass
            c = cz.Context(0)
            if prepass:
                c.set_chain_arena(int(b.length.sum()) * 8 + (64 << 20), min_sequences=0)
            if prepass == 2:
                c.set_literal_arena(int(b.regen.sum()) + (16 << 20))
            t = time.time()
            got = cz.decode_batch_host(frames, caps, c)
            bad = 0
            for i, (r, out) in enumerate(got):
                if int(r["status"]) != int(ost[i]) or (int(ost[i]) == 0 and out != ref_out[int(o_off[i]): int(o_off[i]) + int(olen[i])].tobytes()):
                    bad += 1
                    if bad <= 3:
                        print(f"   frame {i}: gpu {cz.status.name(r['status'])} oracle {cz.status.name(int(ost[i]))}", flush=True)
            print(f"{kind + ('!' if damaged else ''):14s} n={n} prepass={prepass} bad={bad} chain_ms={c.last_chain_ms():.2f} ({time.time() - t:.1f} s)", flush=True)
            total_bad += bad
            c.close()
    print("TOTAL BAD", total_bad)
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
