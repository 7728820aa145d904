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
                        force: bool = False, return_flags: bool = False):
    """Sum `p.grad` over the ranks of `group` in flat buckets of ~bucket_bytes (xGMI is point to point:
    a ring all-reduce is per-link bound, so few large messages; the 141 MB of fp32 gradients are 3 buckets).
    Every rank must issue the same collectives, so a parameter without a gradient on this rank contributes
    zeros -- but a parameter that received no gradient on ANY rank keeps `grad = None`, exactly as the
    single-process reference step leaves it (AdamW skips such parameters: no weight decay, no moment update;
    `training/trainer.py:344-362`).  `force`: issue the collectives at world size 1 too (tests).
    `return_flags`: return the per-parameter "has a gradient on some rank" list (None when nothing was reduced)."""
    if not dist.is_available() or not dist.is_initialized():
        return None
    if dist.get_world_size(group) == 1 and not force:
        return None
    params = [p for p in params if p.requires_grad]
    if not params:
        return None
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
    return anywhere if return_flags else None


class OverlappedGradReducer:
    """Sum of the parameter gradients over the ranks, OVERLAPPED with the backward pass that produces them (the seam is
    `training/trainer.py:344-362`: stage 3 back-propagates minibatch by minibatch, the optimizer steps afterwards; the
    reference is single-process and has no reduction at all).

    Design for xGMI (point-to-point links: a ring all-reduce is per-link bound, so few, large messages):
      * static buckets of ~bucket_bytes over the parameters in REVERSE registration order (the order their gradients become
        final in the backward); a bucket is flattened and all-reduced asynchronously on a side stream as soon as every used
        parameter in it has its final gradient (`register_post_accumulate_grad_hook`), while the backward continues;
      * buckets are launched strictly IN ORDER on every rank, whatever the timing, so the sequence of collectives is the
        same everywhere by construction; what is still missing at `finish()` goes out then, in the same order -- also on a
        rank that never called `arm()` in this step (no local minibatch): it sends every bucket from `finish()`;
      * which parameters are "used" (receive a gradient on some rank) is not knowable before a backward: the first step of a
        reducer runs the non-overlapped `allreduce_gradients` and records the all-reduced has-gradient flags; later steps
        assume that set, contribute zeros for a used parameter that has no gradient on this rank, and all-reduce the REAL
        has-gradient flags of ALL parameters at `finish()` (one small collective): a used parameter that got no gradient
        on any rank this step goes back to `grad = None` (AdamW then skips it, as the reference's step does), a parameter
        outside the set that did get one is reduced on the spot and joins the set.
    Only the LAST backward of a step may launch (`arm()` before it): earlier minibatches only accumulate.
    `force`: run the collectives at world size 1 too (single-GPU RCCL test)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, bucket_bytes: int = 64 << 20, force: bool = False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.force = force
        self.used = None                       # per parameter: receives a gradient on some rank (learnt on the first step)
        self.buckets: List[List[int]] = []
        self.armed = False
        self._handles = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(self.params)]
        self._stream = None
        self._works = []
        self.launched_during_backward = 0      # diagnostics of the last step

    # -- bookkeeping -------------------------------------------------------------------------
    def _active(self) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return self.force or dist.get_world_size(self.group) > 1

    def _build_buckets(self):
        self.buckets, cur, size = [], [], 0
        for i in reversed(range(len(self.params))):
            if not self.used[i]:
                continue
            cur.append(i)
            size += self.params[i].numel() * self.params[i].element_size()
            if size >= self.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)

    def _make_hook(self, i):
        def hook(_p):
            if self.armed:
                self._ready[i] = True
                self._launch_ready()
        return hook

    def _begin(self):
        self._ready = [False] * len(self.params)
        self._had_grad = [False] * len(self.params)     # real (not zero-filled) gradient on THIS rank
        self._next = 0
        self._works = []
        self.launched_during_backward = 0

    def arm(self):
        """Call right before the LAST backward of the step: from here on a parameter's gradient is final when its hook fires."""
        if not self._active() or self.used is None:
            return
        self._begin()
        self.armed = True

    def reset(self):
        """Drop the state of a step that did not reach `finish()` (exception in the backward).  Collectives already in
        flight are waited for so that their buffers may be freed; the gradients of that step are not meaningful."""
        self.armed = False
        for work, _flat, _grads in self._works:
            try:
                work.wait()
            except Exception:                          # pragma: no cover
                pass
        self._works = []

    # -- launching ---------------------------------------------------------------------------
    def _launch_ready(self):
        while self._next < len(self.buckets) and all(self._ready[i] for i in self.buckets[self._next]):
            self._launch(self._next)
            self._next += 1
            self.launched_during_backward += 1

    def _launch(self, b):
        idx = self.buckets[b]
        dev = self.params[idx[0]].device
        grads = []
        for i in idx:
            p = self.params[i]
            if p.grad is None:                 # used elsewhere, untouched here: zeros
                p.grad = torch.zeros_like(p)
            else:
                self._had_grad[i] = True
            grads.append(p.grad)
        if dev.type == 'cuda':
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            self._stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._stream):
                flat = torch.cat([g.reshape(-1) for g in grads])
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            flat = torch.cat([g.reshape(-1) for g in grads])
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, flat, grads))

    def finish(self):
        """After the last backward: send what is left (same order on every rank), wait, write the sums back, verify the set."""
        if not self._active():
            self.armed = False
            return
        if self.used is None:                  # first step: learn the set with the non-overlapped reduction
            has = allreduce_gradients(self.params, self.group, self.bucket_bytes, force=self.force, return_flags=True)
            self.used = has
            self._build_buckets()
            return
        if not self.armed:                     # arm() was not called on this rank: the SAME bucket sequence, all from here
            self._begin()
        self.armed = False
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1
        dev = self.params[0].device
        cuda = dev.type == 'cuda'
        # The copy-back must be ordered after the collective ON THE STREAM IT RUNS ON: for the RCCL process group
        # `work.wait()` only makes the *current* stream wait for the collective's own stream (it does not block the
        # host), so it has to be called with the side stream current -- the stream `flat` was produced on and the
        # copy-back is enqueued on.  (gloo's wait() blocks the host; the same code is right there too.)
        for work, flat, grads in self._works:
            ctx = torch.cuda.stream(self._stream) if cuda else _null()
            with ctx:
                work.wait()
                off = 0
                for g in grads:
                    g.copy_(flat[off:off + g.numel()].view_as(g))
                    off += g.numel()
                if cuda:
                    for g in grads:            # the gradients are consumed on the main stream afterwards
                        g.record_stream(self._stream)
        if cuda:
            torch.cuda.current_stream(dev).wait_stream(self._stream)
        self._works = []
        # the REAL has-gradient flags of every parameter, summed over the ranks (one small collective)
        has = torch.tensor([1.0 if (self._had_grad[i] if u else self.params[i].grad is not None) else 0.0
                            for i, u in enumerate(self.used)], dtype=torch.float32, device=dev)
        dist.all_reduce(has, op=dist.ReduceOp.SUM, group=self.group)
        anywhere = (has > 0).tolist()
        late = []
        for i, (u, a) in enumerate(zip(self.used, anywhere)):
            if u and not a:
                self.params[i].grad = None     # used before, no gradient on ANY rank this step: as the reference leaves it
            elif a and not u:
                late.append(i)
        if late:                               # the set grew: reduce the newcomers now (same list on every rank)
            allreduce_gradients([self.params[i] for i in late], self.group, self.bucket_bytes, force=self.force)
            for i in late:
                self.used[i] = True
            self._build_buckets()

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def multistaged_training_step(model: torch.nn.Module, minibatches: List[dict], positives_mask: torch.Tensor,
                              negatives_mask: torch.Tensor, loss_fn: Callable, optimizer=None,
                              phase: str = 'train', n_total: Optional[int] = None, group=None,
                              force_collectives: bool = False, reducer: Optional[OverlappedGradReducer] = None) -> dict:
    """One step.  `minibatches`: this rank's batch dicts ({'octree': ...}, already on the device with
    neighbours built), in global order; masks are (B_total, B_total) over the whole batch.  Returns the
    loss statistics (identical on every rank).  `n_total`: global batch size (default B_local * world).
    `force_collectives`: issue the all-gather and the gradient all-reduce at world size 1 too (single-GPU RCCL test).
    `reducer`: an `OverlappedGradReducer` over model.parameters() kept across steps: the gradient all-reduce then runs bucket
    by bucket behind the last minibatch's backward instead of after it."""
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
    try:
        for k, mb in enumerate(minibatches):
            y = model(mb)['global']
            if reducer is not None and k == len(minibatches) - 1:
                reducer.arm()                              # gradients are final from here on: buckets may leave
            y.backward(gradient=grad_local[i:i + y.shape[0]])
            i += y.shape[0]
    except BaseException:
        if reducer is not None:
            reducer.reset()                                # no stale armed state / in-flight work for the next step
        raise
    if reducer is not None:
        reducer.finish()
    else:
        allreduce_gradients(model.parameters(), group, force=force_collectives)
    if optimizer is not None:
        optimizer.step()
    return stats
