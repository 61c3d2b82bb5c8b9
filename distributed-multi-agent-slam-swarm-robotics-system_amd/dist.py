"""Multi-GPU: shard telemetry by agent, fuse per-GPU grids with RCCL over xGMI.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).  Every rank raycasts its
own bots' packets into a full-size local stamp grid; because a stamp carries the packet's GLOBAL
arrival index, the cell-wise MAX of the per-rank grids is bit-identical to one mapper fed the
interleaved stream (shared-grid semantics of dual_bot_mapper.py:785).  Hit/miss counters add.
torch is plumbing here: device memory views and the collective, no arithmetic.
"""
import numpy as np


class _DevArray:
    """Minimal __cuda_array_interface__ holder so torch can alias library-owned device memory."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": shape, "typestr": typestr,
                                         "version": 2, "strides": None}


def grid_tensors(mapper, device):
    """(stamps int32 [size,size], counts int32 [size,size,2] or None) aliasing the context's
    device buffers.  Stamps stay below 2^31 by construction, so int32 MAX is exact."""
    import torch
    sp, sb, cp, cb = mapper.device_buffers()
    n = mapper.size
    stamps = torch.as_tensor(_DevArray(sp, (n, n), "<i4"), device=device)
    counts = torch.as_tensor(_DevArray(cp, (n, n, 2), "<i4"), device=device) if cp else None
    return stamps, counts


def allreduce_tensors(stamps, counts=None, group=None):
    """The fuse rule on plain tensors (any backend): latest stamp wins, counters add."""
    import torch.distributed as dist
    dist.all_reduce(stamps, op=dist.ReduceOp.MAX, group=group)
    if counts is not None:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)


def allreduce_grids(mapper, device, group=None, counts=True):
    """Fuse every rank's grid into the global map, in place on all ranks (RCCL over xGMI)."""
    stamps, cnt = grid_tensors(mapper, device)
    allreduce_tensors(stamps, cnt if counts else None, group)


def tri_state_from_stamps(stamps):
    """Host decode of a stamp grid to OccupancyGrid values (-1 / 0 / 100), for checks."""
    s = np.asarray(stamps).astype(np.int64)
    return np.where(s == 0, -1, np.where(s & 1, 100, 0)).astype(np.int8)


def shard_of_bot(bot_index, bots_per_gpu):
    """global bot index (0-based) -> (rank, agent_id on that rank's wire, 1-based)."""
    return bot_index // bots_per_gpu, bot_index % bots_per_gpu + 1


def rank_sequence(rank, world, step_base, n):
    """Global arrival indices of rank `rank`'s n records when `world` equal streams are
    interleaved round-robin: seq = step_base + i*world + rank."""
    return step_base + np.arange(n, dtype=np.uint64) * np.uint64(world) + np.uint64(rank)
