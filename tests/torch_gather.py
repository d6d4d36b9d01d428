"""torch.distributed stand-ins of the two collectives of footprint_tools_amd.distributed (TEST
infrastructure: the product binds RCCL itself and imports no torch).  They move the ranks' track
slices over a gloo process group so that the host logic around the collective -- sharding by padded
bases, shard sizes and offsets (`distributed.shard_offsets`), the sharded driver -- runs with two
ranks on CPU (tests/test_distributed_cpu.py)."""
import numpy as np


def allgather_track(local, sizes, group=None):
    """torch.distributed form of the same collective, for hosts that already run a process group
    (and for the CPU test of the host logic with the gloo backend): all-gather the ranks' track
    slices into the whole track.  Ragged slices are padded to the longest one."""
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


def gather_track(local, sizes, root=0, group=None):
    """the root-gather (fpt_gather_track): the whole track on `root`, None on the other ranks; the slices
    land at footprint_tools_amd.distributed.shard_offsets(sizes)"""
    import torch
    import torch.distributed as dist

    from footprint_tools_amd.distributed import shard_offsets

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [int(s) for s in sizes]
    if len(sizes) != world or local.numel() != sizes[rank]:
        raise ValueError("slice sizes do not fit the ranks")
    off = shard_offsets(sizes)
    m = max(sizes)
    padded = torch.zeros(m, dtype=local.dtype)
    padded[:sizes[rank]] = local
    bufs = [torch.empty(m, dtype=local.dtype) for _ in range(world)] if rank == root else None
    dist.gather(padded, bufs, dst=root, group=group)
    if rank != root:
        return None
    full = torch.empty(int(off[-1]), dtype=local.dtype)
    for r in range(world):
        full[int(off[r]):int(off[r + 1])] = bufs[r][:sizes[r]]
    return full
