"""Multi-GPU plumbing: frames shard by rank (frames share nothing — FrameDecoderState::reset,
src/frame_decoder.cairo:78-104), no data-path collective; the only exchange is the optional
final gather of the decoded arenas, done as direct peer sends to the root so that all of its
xGMI links carry traffic at once (a ring would be bound by one link).

Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on the GPUs, "gloo" in the
CPU tests."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_first_index(frames_per_rank: int, rank: int) -> int:
    """Rank r decodes frames [r*F, (r+1)*F) of the global batch (weak scaling: F fixed per GPU)."""
    return rank * frames_per_rank


def max_over_ranks(seconds: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def all_sizes(nbytes: int, device):
    world = dist.get_world_size()
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([nbytes], dtype=torch.int64, device=device))
    return [int(s.item()) for s in sizes]


def gather_to_root(local: torch.Tensor, recv_bufs, root: int = 0):
    """Variable-size gather of the decoded arenas.  recv_bufs (root only) = one uint8 tensor per
    rank, sized from all_sizes(); the root's own slot is not touched (its arena stays in place)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1:
        return
    if rank == root:
        ops = [dist.P2POp(dist.irecv, recv_bufs[r], r) for r in range(world) if r != root]
    else:
        ops = [dist.P2POp(dist.isend, local, root)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()


def partition_balanced(weights, world: int):
    """Deals items to `world` ranks so that the sums of `weights` are balanced (sort by expected bytes,
    deal greedily — longest processing time first).  Deterministic: every rank computes the same assignment.
    Returns an int array: rank of every item."""
    import heapq
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    order = np.lexsort((np.arange(w.size), -w))          # heaviest first, index breaks ties
    heap = [(0.0, r) for r in range(world)]
    assign = np.empty(w.size, dtype=np.int64)
    for i in order:
        load, r = heapq.heappop(heap)
        assign[i] = r
        heapq.heappush(heap, (load + float(w[i]), r))
    return assign


def rebalance_frames(base, off, length, regen, device):
    """Moves frames between ranks so that every rank holds a byte-balanced share of the GLOBAL batch (each rank
    starts with a contiguous slice).  The weight of a frame is its algorithmic bytes (compressed + decoded).
    One exchange of the compressed bytes (all_to_all_single: RCCL on the GPUs, gloo in the CPU tests), before any
    timing.  Returns (base, off, length, regen, global_index) of the frames this rank now owns; the same bytes as a tensor on
    `device` are left in rebalance_frames.last_device_base."""
    import numpy as np
    rank, world = dist.get_rank(), dist.get_world_size()
    n = int(length.size)
    meta = torch.tensor(np.stack([length.astype(np.int64), regen.astype(np.int64)]), dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(gathered, meta)
    g = np.stack([t.cpu().numpy() for t in gathered])                  # [world, 2, n]
    glen, gregen = g[:, 0, :].reshape(-1), g[:, 1, :].reshape(-1)
    assign = partition_balanced(glen + gregen, world)
    mine = np.arange(rank * n, (rank + 1) * n)
    send_order = [mine[assign[mine] == q] for q in range(world)]       # my frames, grouped by destination, index order
    recv_order = [np.arange(q * n, (q + 1) * n)[assign[q * n:(q + 1) * n] == rank] for q in range(world)]
    sbuf = np.concatenate([base[int(off[i - rank * n]): int(off[i - rank * n] + length[i - rank * n])] for grp in send_order for i in grp]
                          or [np.zeros(0, np.uint8)])
    in_split = [int(glen[grp].sum()) for grp in send_order]
    out_split = [int(glen[grp].sum()) for grp in recv_order]
    t_in = torch.from_numpy(np.ascontiguousarray(sbuf)).to(device)
    t_out = torch.empty(sum(out_split), dtype=torch.uint8, device=device)
    dist.all_to_all_single(t_out, t_in, out_split, in_split)
    idx = np.concatenate(recv_order) if recv_order else np.zeros(0, np.int64)
    nlen = glen[idx].astype(np.uint64)
    noff = np.zeros(idx.size, dtype=np.uint64)
    if idx.size > 1:
        noff[1:] = np.cumsum(nlen[:-1])
    # The exchanged bytes stay where the exchange left them: on the GPU under RCCL (dev_base: what the decode reads — no trip through
    # the host in the data path).  The host copy is for the caller's CPU-side checks only.
    dev_base = torch.cat([t_out, torch.zeros(64, dtype=torch.uint8, device=t_out.device)])
    nbase = dev_base.cpu().numpy()
    rebalance_frames.last_device_base = dev_base
    return nbase, noff, nlen, gregen[idx].astype(np.uint64), idx


rebalance_frames.last_device_base = None
