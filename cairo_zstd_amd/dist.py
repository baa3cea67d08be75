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
