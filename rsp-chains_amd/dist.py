"""Multi-GPU plumbing: one process per GPU, chirps / Rx channels sharded across ranks with
no data-path collective; the only exchange is the all-gather of the final detection lists
(RCCL over xGMI when the tensors live on GPUs; the same code runs on gloo/CPU tensors, which
is how the N > 1 path is tested without GPUs).

The reference has no counterpart: its chain is one stream, one clock domain
(/root/reference/src/main/scala/FftMagCfarChain.scala:47).  Frames are independent there too,
which is what makes the sharding embarrassingly parallel.
"""
from __future__ import annotations

from typing import Tuple

DET_WORDS = 4  # rsp_detection = {frame, bin, doppler, word}, 4 x uint32


def shard_range(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of units (frames / channels) owned by `rank`: sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError(f"requirement failed: rank {rank} of {world}")
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_detections(local_list, local_count, cap: int, frame_offset: int = 0, group=None,
                      out_list=None, out_counts=None, async_op: bool = False):
    """All-gather fixed-capacity detection lists.

    local_list : int32 tensor [cap, 4] (rows beyond local_count are ignored)
    local_count: int32 tensor [1]
    Returns (lists [world, cap, 4], counts [world]) -- or, with async_op, the two work handles too.
    Frame indices stay rank-local; add shard_range(...)[0] (or pass frame_offset to
    merge_gathered) to make them global.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if out_list is None:
        out_list = torch.empty((world * cap, DET_WORDS), dtype=local_list.dtype, device=local_list.device)
    if out_counts is None:
        out_counts = torch.empty((world,), dtype=local_count.dtype, device=local_count.device)
    h1 = dist.all_gather_into_tensor(out_counts, local_count, group=group, async_op=async_op)
    h2 = dist.all_gather_into_tensor(out_list, local_list[:cap].reshape(cap, DET_WORDS), group=group,
                                     async_op=async_op)
    lists = out_list.view(world, cap, DET_WORDS)
    if async_op:
        return lists, out_counts, (h1, h2)
    return lists, out_counts


def gather_packed(packed, group=None, out=None, async_op: bool = False):
    """ONE collective per step: `packed` is an int32 tensor [cap + 1, 4] whose row 0 is the header
    {peaks found, entries stored, -, -} the C ABI writes through d_count, and rows 1.. the list (point
    d_count at row 0 and d_list at row 1).  Returns the gathered [world, cap + 1, 4] tensor (and the
    work handle with async_op)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rows = packed.shape[0]
    if out is None:
        out = torch.empty((world * rows, DET_WORDS), dtype=packed.dtype, device=packed.device)
    h = dist.all_gather_into_tensor(out, packed, group=group, async_op=async_op)
    view = out.view(world, rows, DET_WORDS)
    return (view, h) if async_op else view


def unpack_gathered(view):
    """[world, cap + 1, 4] from gather_packed -> (lists [world, cap, 4], stored [world], found [world]).
    Only the first stored[r] rows of lists[r] are valid (found[r] > stored[r] = that rank's list was truncated)."""
    return view[:, 1:, :], view[:, 0, 1], view[:, 0, 0]


def merge_gathered(lists, counts, frames_per_rank):
    """Host-side: concatenate the valid rows of every rank with global frame numbers.
    counts[r] = entries STORED by rank r (the header's second word), frames_per_rank[r] = first global
    frame of rank r."""
    import numpy as np
    lists = lists.cpu().numpy() if hasattr(lists, "cpu") else np.asarray(lists)
    counts = counts.cpu().numpy() if hasattr(counts, "cpu") else np.asarray(counts)
    out = []
    for r in range(lists.shape[0]):
        k = min(int(counts[r]), lists.shape[1])
        part = lists[r, :k].astype(np.int64).copy()
        part[:, 0] += int(frames_per_rank[r])
        out.append(part)
    return np.concatenate(out, axis=0) if out else np.zeros((0, DET_WORDS), np.int64)
