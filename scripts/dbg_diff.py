#!/usr/bin/env python3
"""Diagnostic: decode a small synthetic batch with the library named by CAIRO_ZSTD_AMD_LIB and print where frames differ from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cairo_zstd_amd as cz
import oracle
from cairo_zstd_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "full_4a"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
b = synth.generate(kind, n, nthreads=8)
frames = [b.frame(i) for i in range(n)]
caps = [int(r) + 8 for r in b.regen]
ctx = cz.Context(0)
ctx.set_chain_arena(256 << 20, min_sequences=0)
ctx.set_literal_arena(128 << 20)
got = cz.decode_batch_host(frames, caps, ctx)
nbad = 0
for i, (fr, cap, (r, out)) in enumerate(zip(frames, caps, got)):
    st, ref, info = oracle.decode_frame(fr, cap=cap)
    if int(r["status"]) != st or out != ref:
        nbad += 1
        if nbad <= 4:
            m = min(len(out), len(ref))
            d = [k for k in range(m) if out[k] != ref[k]]
            print(f"frame {i}: status {int(r['status'])} vs {st}, len {len(out)} vs {len(ref)}, {len(d)} bytes differ, first {d[:12]}")
            if d:
                k = d[0]
                print("  got", out[max(0, k - 8):k + 24].hex(), "\n  ref", ref[max(0, k - 8):k + 24].hex())
                runs = []
                s = d[0]; p = d[0]
                for q in d[1:]:
                    if q != p + 1:
                        runs.append((s, p - s + 1)); s = q
                    p = q
                runs.append((s, p - s + 1))
                print("  runs (start, len):", runs[:16])
print("bad frames:", nbad, "of", n)
ctx.close()
