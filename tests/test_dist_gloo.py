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

pytestmark = pytest.mark.xdist_group("gloo")        # (one worker: the rendezvous tests do not share a box's ports and cores with each other)
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


def test_partition_balanced_is_deterministic_and_balanced():
    rng = np.random.default_rng(5)
    w = np.concatenate([rng.integers(1, 300, 900), rng.integers(50_000, 900_000, 100)])    # a corpus-like long tail
    a = czdist.partition_balanced(w, 8)
    assert (a == czdist.partition_balanced(w, 8)).all() and a.min() == 0 and a.max() == 7
    loads = np.array([w[a == r].sum() for r in range(8)], dtype=np.float64)
    assert loads.max() / loads.mean() < 1.02
    contiguous = np.array([w[r * 125:(r + 1) * 125].sum() for r in range(8)], dtype=np.float64)
    assert loads.max() <= contiguous.max()


def _rebalance_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = synth.generate("mix", F, first_index=czdist.shard_first_index(F, rank), nthreads=1)
    nb, noff, nlen, nregen, idx = czdist.rebalance_frames(batch.base, batch.off, batch.length, batch.regen, torch.device("cpu"))
    frames = [hashlib.sha256(nb[int(o): int(o + l)].tobytes()).hexdigest() for o, l in zip(noff, nlen)]
    q.put((rank, idx.tolist(), frames, [int(x) for x in nregen], float((nlen + nregen).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rebalance_by_algorithmic_bytes():
    """BASELINE config 5's dealing step: after the exchange every frame of the global batch lives on exactly one rank,
    byte for byte, and the two loads are balanced."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rebalance_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = synth.generate("mix", 2 * F, nthreads=1)
    seen = sorted(msgs[0][1] + msgs[1][1])
    assert seen == list(range(2 * F))
    for rank, idx, frames, regen, load in msgs:
        for i, h, rg in zip(idx, frames, regen):
            assert hashlib.sha256(whole.frame(i)).hexdigest() == h and int(whole.regen[i]) == rg
    loads = [m[4] for m in msgs]
    heaviest = float((whole.length + whole.regen).max())
    assert abs(loads[0] - loads[1]) <= heaviest


def test_bench_gpus_flag_fails_loudly_without_the_gpus():
    """python bench.py --gpus N must start N ranks or fail: it never runs one rank and reports it as N."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "needs 2 GPUs" in p.stderr and "n_gpus" not in p.stdout


def _world8_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    f8 = 6
    batch = synth.generate("mix", f8, first_index=czdist.shard_first_index(f8, rank), nthreads=1)
    nb, noff, nlen, nregen, idx = czdist.rebalance_frames(batch.base, batch.off, batch.length, batch.regen, dev)
    nbatch = synth.Batch(nb, noff, nlen, nregen)
    arena = torch.from_numpy(_decode_shard(nbatch)) if nbatch.n else torch.zeros(0, dtype=torch.uint8)
    sizes = czdist.all_sizes(arena.numel(), dev)
    bufs = [torch.empty(s, dtype=torch.uint8) for s in sizes] if rank == 0 else None
    czdist.gather_to_root(arena, bufs, 0)
    out_off, _, _ = nbatch.out_layout(64) if nbatch.n else (np.zeros(0, np.uint64), None, 0)
    mine = {int(i): hashlib.sha256(arena.numpy()[int(o): int(o + r)].tobytes()).hexdigest() for i, o, r in zip(idx, out_off, nregen)}
    if rank == 0:
        gathered = [hashlib.sha256(arena.numpy().tobytes()).hexdigest()] + [hashlib.sha256(bufs[r].numpy().tobytes()).hexdigest() for r in range(1, world)]
        q.put((rank, mine, hashlib.sha256(arena.numpy().tobytes()).hexdigest(), float((nlen + nregen).sum()), gathered))
    else:
        q.put((rank, mine, hashlib.sha256(arena.numpy().tobytes()).hexdigest(), float((nlen + nregen).sum()), None))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_8_deal_decode_and_gather():
    """BASELINE config 5 rehearsed at its real world size on the CPU (gloo): 8 ranks x 6 corpus-like frames — deal by algorithmic
    bytes (partition_balanced + one all_to_all), decode every share (the oracle stands in for the GPU here), gather the eight
    decoded arenas to rank 0 by direct peer sends.  Every frame of the global batch is decoded exactly once and correctly, the
    loads are balanced, and rank 0 ends up with every rank's arena bit for bit."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda m: m[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    whole = synth.generate("mix", 8 * 6, nthreads=1)
    seen = {}
    for rank, mine, arena_hash, load, gathered in msgs:
        for i, h in mine.items():
            assert i not in seen
            seen[i] = h
    assert sorted(seen) == list(range(48))
    for i, h in seen.items():
        st, ref, _ = oracle.decode_frame(whole.frame(i), cap=int(whole.regen[i]) + 64)
        assert st == 0 and hashlib.sha256(ref).hexdigest() == h, i
    root = msgs[0]
    assert root[4] == [m[2] for m in msgs]                               # rank 0 holds every rank's decoded arena
    loads = np.array([m[3] for m in msgs])
    assert loads.max() - loads.min() <= float((whole.length + whole.regen).max())
