"""Multi-staged training step of the reference (`training/trainer.py:287-365`, after "Learning with Average
Precision: Training Image Retrieval with a Listwise Loss") as a data-parallel step over RCCL:

  stage 1  every rank encodes its own contiguous, ordered slice of the global batch, minibatch by minibatch,
           without autograd (`trainer.py:305-317`);
  stage 2  the (B_local, D) descriptor blocks are all-gathered in rank order, every rank evaluates the listwise
           loss on the full (B_total, D) matrix and keeps d loss / d embeddings (`trainer.py:321-338`);
  stage 3  every rank re-encodes its minibatches with autograd and back-propagates its rows of that gradient
           (`trainer.py:344-358`); parameter gradients are then summed across ranks with a bucketed all-reduce
           (the loss is one global scalar, so the sum of the per-rank partial gradients IS its gradient) and the
           optimizer steps.

The reference is single-process; with world size 1 (or no process group) this is its step verbatim.
Descriptors depend on the ordered sub-batch (windows straddle clouds), so parity with a single-process run
holds for the same minibatch partition, not for a re-partitioned batch."""

from typing import Callable, Iterable, List, Optional

import torch
import torch.distributed as dist

from .distributed import all_gather_descriptors


def _stage1_numerics(model, phase):
    """Stage 1 must reproduce what stage 3 recomputes (see hotformerloc_amd.model.training_numerics); models
    that are not this package's encoder (tests) need nothing."""
    import contextlib
    if phase != 'train':
        return contextlib.nullcontext()
    try:
        from .model import training_numerics
    except Exception:                                   # pragma: no cover
        return contextlib.nullcontext()
    return training_numerics()


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20,
                        force: bool = False):
    """Sum `p.grad` over the ranks of `group` in flat buckets of ~bucket_bytes (xGMI is point to point:
    a ring all-reduce is per-link bound, so few large messages; the 141 MB of fp32 gradients are 3 buckets).
    Every rank must issue the same collectives, so a parameter without a gradient on this rank contributes
    zeros -- but a parameter that received no gradient on ANY rank keeps `grad = None`, exactly as the
    single-process reference step leaves it (AdamW skips such parameters: no weight decay, no moment update;
    `training/trainer.py:344-362`).  `force`: issue the collectives at world size 1 too (tests)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    if dist.get_world_size(group) == 1 and not force:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    dev = params[0].device
    # which parameters have a gradient somewhere: one small all-reduce of a flag vector
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32, device=dev)
    dist.all_reduce(has, op=dist.ReduceOp.SUM, group=group)
    anywhere = (has > 0).tolist()
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        bucket, size = [], 0

    for p, used in zip(params, anywhere):
        if not used:
            continue                                   # None on every rank: stays None
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        bucket.append(p.grad)
        size += p.grad.numel() * p.grad.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def multistaged_training_step(model: torch.nn.Module, minibatches: List[dict], positives_mask: torch.Tensor,
                              negatives_mask: torch.Tensor, loss_fn: Callable, optimizer=None,
                              phase: str = 'train', n_total: Optional[int] = None, group=None,
                              force_collectives: bool = False) -> dict:
    """One step.  `minibatches`: this rank's batch dicts ({'octree': ...}, already on the device with
    neighbours built), in global order; masks are (B_total, B_total) over the whole batch.  Returns the
    loss statistics (identical on every rank).  `n_total`: global batch size (default B_local * world).
    `force_collectives`: issue the all-gather and the gradient all-reduce at world size 1 too (single-GPU RCCL test)."""
    assert phase in ('train', 'val')
    model.train() if phase == 'train' else model.eval()
    # ---- stage 1 ------------------------------------------------------------------------------
    with torch.no_grad(), _stage1_numerics(model, phase):
        local = torch.cat([model(mb)['global'] for mb in minibatches], 0)
    embeddings = all_gather_descriptors(local, n_total, group, force=force_collectives).detach()
    # ---- stage 2 ------------------------------------------------------------------------------
    with torch.set_grad_enabled(phase == 'train'):
        if phase == 'train':
            embeddings.requires_grad_(True)
        loss, stats = loss_fn(embeddings, positives_mask, negatives_mask)
        if phase == 'train':
            loss.backward()
    if phase != 'train':
        return stats
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if world > 1:
        from .distributed import shard_bounds
        lo = shard_bounds(embeddings.shape[0], rank, world)[0]
    else:
        lo = 0
    grad_local = embeddings.grad[lo:lo + local.shape[0]]
    # ---- stage 3 ------------------------------------------------------------------------------
    if optimizer is not None:
        optimizer.zero_grad()
    else:
        model.zero_grad(set_to_none=True)
    i = 0
    for mb in minibatches:
        y = model(mb)['global']
        y.backward(gradient=grad_local[i:i + y.shape[0]])
        i += y.shape[0]
    allreduce_gradients(model.parameters(), group, force=force_collectives)
    if optimizer is not None:
        optimizer.step()
    return stats
