#!/usr/bin/env python3
"""Diagnostic: 300 000 tiny frames (Raw frames of 0..29 bytes, every seventh a small corpus frame) in one batch through the
pre-pass pipeline — lists and arenas overflow by design, every frame must still come out right."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cairo_zstd_amd as cz, oracle
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import raw_frame_with_checksum, corpus_pairs
rng = np.random.default_rng(5)
small = [(z, orig) for name, z, orig in corpus_pairs(max_orig=300)]
frames, caps, refs = [], [], []
for i in range(300000):
    if i % 7 == 0:
        z, orig = small[i % len(small)]
        frames.append(z); caps.append(len(orig) + 8); refs.append(orig)
    else:
        n = int(rng.integers(0, 30)); b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        frames.append(raw_frame_with_checksum(b)); caps.append(n + 4); refs.append(b)
c = cz.Context(0)
c.set_chain_arena(64 << 20, min_sequences=0); c.set_literal_arena(32 << 20)
got = cz.decode_batch_host(frames, caps, c)
bad = sum(1 for (r, out), ref in zip(got, refs) if int(r["status"]) != 0 or out != ref)
print("frames", len(frames), "bad", bad, "kernel ms", c.last_kernel_ms(), "prepass counts", c.last_prepass_counts(len(frames)))
c.close()
