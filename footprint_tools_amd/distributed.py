"""Interval-sharded scans across the GPUs of one node (one process per GPU).

Intervals are independent (every window is confined to its own padded interval), so a batch is
split into contiguous interval ranges balanced by padded bases (`scan.shard_intervals`), every
rank scans its own range with no communication, and the per-base statistics track is
re-assembled on every rank with ONE collective at the end: an all-gather over RCCL/xGMI
(`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" works for CPU tensors in tests).
PyTorch is only plumbing here (process group, device tensors); the scan itself is libfpt_hip.
"""
import numpy as np


def shard_track_sizes(lengths, bounds):
    """bases owned by each rank given interval lengths and [(first, last), ...] ranges."""
    lengths = np.asarray(lengths, dtype=np.int64)
    return [int(lengths[a:b].sum()) for a, b in bounds]


def allgather_track(local, sizes, group=None):
    """All-gather the ranks' track slices into the whole track (returned on every rank).

    local : 1-D torch tensor holding this rank's slice (sizes[rank] elements), on the device
            the process group communicates from (CUDA for nccl/RCCL, CPU for gloo).
    sizes : per-rank slice lengths in rank order.
    Equal slices use all_gather_into_tensor directly; ragged slices are padded to the longest
    one (the equal-count form of the collective) and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [int(s) for s in sizes]
    if len(sizes) != world:
        raise ValueError("need one slice size per rank")
    if local.numel() != sizes[rank]:
        raise ValueError("local slice has %d elements, expected %d" % (local.numel(), sizes[rank]))
    m = max(sizes)
    if min(sizes) == m:
        out = torch.empty(world * m, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros(m, dtype=local.dtype, device=local.device)
    padded[:sizes[rank]] = local
    buf = torch.empty(world * m, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[r * m:r * m + sizes[r]] for r in range(world)])
