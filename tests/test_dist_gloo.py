"""N>1 path on CPU: world_size-2 gloo.  Checks the pieces bench.py's multi-GPU leg is made of:
rank sharding of the synthetic batch (disjoint, covers the global batch), max-over-ranks timing,
sums, and the variable-size gather of decoded arenas to rank 0 by direct peer sends.  The decode
itself is done by the oracle here (no GPU in this test)."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from cairo_zstd_amd import dist as czdist
from cairo_zstd_amd import synth

F = 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _decode_shard(batch):
    out_off, out_cap, total = batch.out_layout(64)
    out, olen, st = oracle.decode_batch(batch.base, batch.off, batch.length, out_off, out_cap, total)
    assert (st == 0).all() and (olen == batch.regen).all()
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    batch = synth.generate("mix", F, first_index=czdist.shard_first_index(F, rank), nthreads=1)
    arena = torch.from_numpy(_decode_shard(batch))
    sizes = czdist.all_sizes(arena.numel(), dev)
    bufs = [torch.empty(s, dtype=torch.uint8) for s in sizes] if rank == 0 else None
    czdist.gather_to_root(arena, bufs, 0)
    tmax = czdist.max_over_ranks(1.0 + rank, dev)
    tot = czdist.sum_over_ranks([float(batch.regen.sum()), 1.0], dev)
    digest = [hashlib.sha256(b"".join(batch.frame(i) for i in range(F))).hexdigest()]
    if rank == 0:
        got = [hashlib.sha256(arena.numpy().tobytes()).hexdigest()] + [hashlib.sha256(bufs[r].numpy().tobytes()).hexdigest() for r in range(1, world)]
        q.put(("root", sizes, got, tmax, tot, digest))
    else:
        q.put(("peer", rank, hashlib.sha256(arena.numpy().tobytes()).hexdigest(), tmax, tot, digest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = next(m for m in msgs if m[0] == "root")
    peer = next(m for m in msgs if m[0] == "peer")
    # gather: rank 0 holds rank 1's arena bit for bit
    assert root[2][1] == peer[2]
    # timing = max over ranks, sums = whole job
    assert root[3] == peer[3] == 2.0
    assert root[4][1] == 2.0
    # shards: rank r holds frames [r*F, (r+1)*F) of the single-process batch
    whole = synth.generate("mix", 2 * F, nthreads=1)
    for r, m in ((0, root), (1, peer)):
        want = hashlib.sha256(b"".join(whole.frame(i) for i in range(r * F, (r + 1) * F))).hexdigest()
        assert m[5][0] == want
    assert root[4][0] == float(whole.regen.sum())
