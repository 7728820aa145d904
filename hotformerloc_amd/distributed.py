"""Data-parallel sharding of a batch of clouds and the one exchange step of the path: the
all-gather of global descriptors for the listwise loss (RCCL over xGMI; `backend="nccl"` IS RCCL
on ROCm).  The reference has no distributed code at all; its multi-staged
training step already evaluates the model on contiguous ordered sub-batches
(`datasets/dataset_utils.py:129-134`, `training/trainer.py:309-317`) and back-propagates each
sub-batch from its slice of `embeddings.grad` (`training/trainer.py:344-362`) -- the seam used here.

Descriptors depend on the ordered sub-batch (windows straddle clouds), so a rank always owns a
CONTIGUOUS slice of the global batch, in order.
"""

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`; the first n_total % world ranks get one more."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_clouds(clouds: List, rank: Optional[int] = None, world: Optional[int] = None) -> List:
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    lo, hi = shard_bounds(len(clouds), rank, world)
    return list(clouds[lo:hi])


class _AllGatherDescriptors(torch.autograd.Function):
    """Forward: concatenate every rank's (B_r, D) block in rank order.  Backward: each rank keeps
    the rows of the gradient that belong to its own block (every rank computes the same loss on
    the gathered matrix, as stage 2 of the reference's multi-staged step does on one device)."""

    @staticmethod
    def forward(ctx, local: torch.Tensor, sizes: Tuple[int, ...], group):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        local = local.contiguous()
        if len(set(sizes)) == 1:
            out = local.new_empty((sizes[0] * world, local.shape[1]))
            dist.all_gather_into_tensor(out, local, group=group)
        else:
            # uneven slices: pad every block to the largest one (single collective, equal
            # message sizes -- what RCCL and gloo both take), then drop the padding rows
            nmax = max(sizes)
            padded = local.new_zeros((nmax, local.shape[1]))
            padded[:local.shape[0]] = local
            buf = local.new_empty((nmax * world, local.shape[1]))
            dist.all_gather_into_tensor(buf, padded, group=group)
            out = torch.cat([buf[r * nmax:r * nmax + n] for r, n in enumerate(sizes)], 0)
        ctx.lo = sum(sizes[:rank])
        ctx.n = sizes[rank]
        return out

    @staticmethod
    def backward(ctx, grad):
        return grad[ctx.lo:ctx.lo + ctx.n].contiguous(), None, None


def all_gather_descriptors(local: torch.Tensor, n_total: Optional[int] = None, group=None,
                           force: bool = False) -> torch.Tensor:
    """(B_local, D) on every rank -> (B_total, D) on every rank, rows in global batch order.
    One small latency-bound collective (32 KiB per rank at B_local=32): a single
    `all_gather_into_tensor` (uneven slices are padded to the largest block)."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    if dist.get_world_size(group) == 1 and not force:     # force: still issue the collective (tests)
        return local
    world = dist.get_world_size(group)
    if n_total is None:
        n_total = local.shape[0] * world
    sizes = tuple(shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0]
                  for r in range(world))
    assert sizes[dist.get_rank(group)] == local.shape[0], 'local block does not match shard_bounds'
    return _AllGatherDescriptors.apply(local, sizes, group)
