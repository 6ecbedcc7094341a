"""Multi-GPU sharding of the block-render path.

The path shards over *independent units* — the weakly-connected components of the render graph
below the mix point (voices of a synth, offline render jobs).  Each rank (one process per GPU,
``torch.distributed`` over RCCL/xGMI) builds and renders only its own units; nothing is exchanged
inside a block.  The single exchange step is the output bus:

* ``reduce_bus``   — sum-reduce of the per-rank buses to rank 0 (RCCL reduce; throughput mode), or
* ``ordered_bus_sum`` — gather + sum in rank order on rank 0 (bit-reproducible partial-sum order,
  used when checking parity), and
* ``gather_outputs`` — gather of per-unit outputs (independent render jobs, BASELINE configs[3]).

Graphs that are one connected component do not shard: run replicas.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of the units rank `rank` owns (sizes differ by at most one)."""
    base, extra = divmod(int(num_units), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_bus(bus: torch.Tensor, dst: int = 0, async_op: bool = False):
    """Sum the per-rank output buses [blocks, channels, frames] onto `dst` (in place on dst)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if dist.get_backend() == "gloo" and bus.is_cuda:
        # gloo has no reduce for device tensors (tests: two ranks sharing one GPU); its all-reduce leaves the same sum on dst
        return dist.all_reduce(bus, op=dist.ReduceOp.SUM, async_op=async_op)
    return dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op)


def ordered_bus_sum(bus: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather every rank's bus on `dst` and add them in rank order 0,1,2,... (float addition is not
    associative; a ring reduce picks its own order). Returns the sum on `dst`, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bus.clone()
    world, rank = dist.get_world_size(), dist.get_rank()
    parts: Optional[List[torch.Tensor]] = [torch.empty_like(bus) for _ in range(world)] if rank == dst else None
    dist.gather(bus, parts, dst=dst)
    if rank != dst:
        return None
    acc = parts[0].clone()
    for p in parts[1:]:
        acc += p
    return acc


def gather_outputs(local: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """Concatenate per-unit outputs [units_r, ...] of all ranks along dim 0 on `dst` (ranks may own
    different unit counts, cf. ``shard_range``)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local.clone()
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()          # gloo gathers host tensors only (tests: two ranks sharing one GPU)
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    most = int(max(int(c.item()) for c in counts))
    padded = torch.zeros((most,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, parts, dst=dst)
    if rank != dst:
        return None
    return torch.cat([p[:int(c.item())] for p, c in zip(parts, counts)], dim=0)
