"""Sample-sharded data parallelism for the Monte-Carlo losses.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo"
in the CPU tests).  Parameters are replicated; rank r owns the contiguous block
of every global batch given by `shard_range`; base noise is indexed by GLOBAL
sample index, so results do not depend on the number of ranks.  A loss needs
exactly ONE collective: a sum all-reduce of its stacked partial sums (a few
float64 values; plus the gradient vector once the backward pass exists).

The reference has no counterpart (single-process, single-device jax.jit:
cnf_ot/mfc/solvers.py:90-97).
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass(frozen=True)
class Shard:
  rank: int = 0
  world: int = 1
  group: Optional[object] = None


def current_shard() -> Shard:
  """The default shard: the torch.distributed world if initialised, else one rank."""
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized():
    return Shard(dist.get_rank(), dist.get_world_size(), None)
  return Shard()


def shard_range(n: int, shard: Shard) -> Tuple[int, int]:
  """(first global index, count) of this rank's contiguous block of n samples;
  blocks differ by at most one sample and tile [0, n)."""
  base, rem = divmod(int(n), shard.world)
  start = shard.rank * base + min(shard.rank, rem)
  return start, base + (1 if shard.rank < rem else 0)


def all_reduce_sums(sums: torch.Tensor, shard: Shard) -> torch.Tensor:
  """The one collective of a loss evaluation: in-place sum of the partial sums."""
  if shard.world > 1:
    import torch.distributed as dist
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=shard.group)
  return sums


def broadcast_params(params, src: int = 0, shard: Optional[Shard] = None) -> None:
  """Replicate the parameters of rank `src` (the multi-GPU init step) and tell the engines: torch.distributed
  collectives write in place WITHOUT bumping the tensor's version counter, so an engine that skips unchanged
  parameters (FlowEngine.load(assume_unchanged=True)) would keep computing with the old ones."""
  shard = shard if shard is not None else current_shard()
  flat = params.flat if hasattr(params, "flat") else params
  if shard.world > 1:
    import torch.distributed as dist
    dist.broadcast(flat, src=src, group=shard.group)
  from .flows import mark_updated
  mark_updated(flat)


def all_reduce_params(params, average: bool = True, shard: Optional[Shard] = None) -> None:
  """Sum (or average) the parameters over the ranks in place, and mark them as written (see broadcast_params)."""
  shard = shard if shard is not None else current_shard()
  flat = params.flat if hasattr(params, "flat") else params
  if shard.world > 1:
    import torch.distributed as dist
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=shard.group)
    if average:
      flat.div_(shard.world)
  from .flows import mark_updated
  mark_updated(flat)
