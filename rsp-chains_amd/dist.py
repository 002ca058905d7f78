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


def gather_detections(local_list, local_count, cap: int, group=None, out_list=None, out_counts=None,
                      async_op: bool = False):
    """All-gather fixed-capacity detection lists (two collectives; gather_packed is the one-collective form).

    local_list : int32 tensor [cap, 4] (rows beyond the stored count are ignored)
    local_count: int32 tensor [2] = {peaks found, entries stored} -- the two words the C ABI writes through d_count
                 (rsp_chain_process_detect_device: pointing d_count at a ONE-word tensor is an out-of-bounds write)
    Returns (lists [world, cap, 4], stored [world], found [world]) -- with async_op the two work handles too.
    Frame indices stay rank-local; pass shard_range(...)[0] per rank to merge_gathered to make them global.
    """
    import torch
    import torch.distributed as dist

    if tuple(local_count.shape) != (2,):
        raise ValueError("requirement failed: local_count must hold the two words {found, stored} of d_count")
    world = dist.get_world_size(group)
    if out_list is None:
        out_list = torch.empty((world * cap, DET_WORDS), dtype=local_list.dtype, device=local_list.device)
    if out_counts is None:
        out_counts = torch.empty((world * 2,), dtype=local_count.dtype, device=local_count.device)
    h1 = dist.all_gather_into_tensor(out_counts, local_count, group=group, async_op=async_op)
    h2 = dist.all_gather_into_tensor(out_list, local_list[:cap].reshape(cap, DET_WORDS), group=group,
                                     async_op=async_op)
    lists, counts = out_list.view(world, cap, DET_WORDS), out_counts.view(world, 2)
    if async_op:
        return lists, counts[:, 1], counts[:, 0], (h1, h2)
    return lists, counts[:, 1], counts[:, 0]


def gather_packed(packed, group=None, out=None, async_op: bool = False, rows: int = 0):
    """ONE collective per step: `packed` is an int32 tensor [cap + 1, 4] whose row 0 is the header
    {peaks found, entries stored, -, -} the C ABI writes through d_count, and rows 1.. the list (point
    d_count at row 0 and d_list at row 1).  rows > 0: only the first `rows` rows travel (header + rows - 1
    list entries) -- the lists are a few thousand entries in a 65 536-entry buffer; gathered_complete() tells
    afterwards whether every rank's stored entries fitted (PackedGatherer keeps the size adapted).
    Returns the gathered [world, rows, 4] tensor (and the work handle with async_op)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rows = min(rows, packed.shape[0]) if rows > 0 else packed.shape[0]
    if out is None:
        out = torch.empty((world * rows, DET_WORDS), dtype=packed.dtype, device=packed.device)
    else:
        out = out.view(-1, DET_WORDS)[:world * rows]
    h = dist.all_gather_into_tensor(out, packed[:rows], group=group, async_op=async_op)
    view = out.view(world, rows, DET_WORDS)
    return (view, h) if async_op else view


def gather_packed_batch(batch, group=None, out=None, stage=None, rows: int = 0):
    """The lists of K steps in ONE collective: `batch` is an int32 tensor [K, cap + 1, 4], each [cap + 1, 4] slice a
    packed list as in gather_packed.  A step of tens of microseconds cannot hide a collective of its own (launch
    latency and the CU share of the RCCL kernel: 12 us per 49-us step, measured); K lists per collective amortise it.
    The first `rows` rows of every slice are packed into `stage` ([K * rows, 4], allocated if None) and gathered.
    Returns the gathered [world, K, rows, 4] tensor."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    k = batch.shape[0]
    rows = min(rows, batch.shape[1]) if rows > 0 else batch.shape[1]
    if k == 1:   # one list: its first rows are contiguous already
        stage = batch[0, :rows]
    else:
        if stage is None:
            stage = torch.empty((k * rows, DET_WORDS), dtype=batch.dtype, device=batch.device)
        else:
            stage = stage.view(-1, DET_WORDS)[:k * rows]
        stage.view(k, rows, DET_WORDS).copy_(batch[:, :rows])
    if out is None:
        out = torch.empty((world * k * rows, DET_WORDS), dtype=batch.dtype, device=batch.device)
    else:
        out = out.view(-1, DET_WORDS)[:world * k * rows]
    dist.all_gather_into_tensor(out, stage, group=group)
    return out.view(world, k, rows, DET_WORDS)


def gathered_complete(view) -> bool:
    """Did every rank's stored entries fit the rows that were gathered?  (Reads the headers: synchronises.)"""
    return int(view[..., 0, 1].max().item()) + 1 <= view.shape[-2]


class PackedGatherer:
    """The detection gather of a stream of steps with the payload sized to the lists, not to their capacity.

    north_star: "RCCL over xGMI used only to gather the final detection list".  A step's list holds `stored`
    entries of a fixed-capacity buffer; the collective carries `rows` rows, rows - 1 >= the largest `stored`
    seen so far with a 25 % margin (a multiple of 1024, so that the size changes rarely).  gather() never reads the
    device; settle() -- off the hot path, e.g. every few steps or when a consumer needs the list -- reads the
    gathered headers, grows `rows` when a list did not fit and says whether the view is complete (a caller
    that needs every step complete repeats the gather of that step with the full capacity)."""

    def __init__(self, cap: int, group=None, min_rows: int = 256):
        self.cap, self.group, self.rows = cap, group, min(min_rows, cap + 1)

    def gather(self, packed, out=None, async_op: bool = False, full: bool = False):
        return gather_packed(packed, self.group, out, async_op, rows=0 if full else self.rows)

    def gather_batch(self, batch, out=None, stage=None, full: bool = False):
        return gather_packed_batch(batch, self.group, out, stage, rows=0 if full else self.rows)

    def settle(self, view) -> bool:
        """view: [world, rows, 4] of gather() or [world, K, rows, 4] of gather_batch()"""
        need = int(view[..., 0, 1].max().item()) + 1
        ok = need <= view.shape[-2]
        want = -(-(need + need // 4) // 1024) * 1024   # 25 % headroom, in steps of 1024 rows
        if want > self.rows:
            self.rows = min(want, self.cap + 1)
        return ok


def unpack_gathered(view):
    """[world, rows, 4] from gather_packed -> (lists [world, rows - 1, 4], stored [world], found [world]).
    Only the first stored[r] rows of lists[r] are valid (found[r] > stored[r] = that rank's list was truncated;
    stored[r] > rows - 1 = the gather was sized too small: gathered_complete)."""
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
