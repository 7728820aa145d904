"""N > 1 path on CPU: world-size-2 gloo processes exercise the contiguous sharding and the
descriptor all-gather (forward order + gradient slicing)."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hotformerloc_amd.distributed import all_gather_descriptors, shard_bounds, shard_clouds


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, result_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        clouds = list(range(n_total))
        mine = shard_clouds(clouds)
        lo, hi = shard_bounds(n_total, rank, world)
        assert mine == clouds[lo:hi]
        # "descriptors": row i of the global batch is [i, i, i, i] scaled by a learnable factor
        w = torch.ones(1, requires_grad=True)
        local = torch.tensor([[float(i)] * 4 for i in mine]) * w
        full = all_gather_descriptors(local, n_total)
        assert full.shape == (n_total, 4)
        assert torch.equal(full.detach()[:, 0], torch.arange(n_total, dtype=torch.float32))
        # a listwise loss on the gathered matrix; each rank gets the gradient of its own rows only
        weights = torch.arange(1, n_total + 1, dtype=torch.float32).unsqueeze(1)
        (full * weights).sum().backward()
        expect = sum(4.0 * i * (i + 1) for i in mine)
        assert abs(w.grad.item() - expect) < 1e-4, (w.grad.item(), expect)
        torch.save(full.detach(), os.path.join(result_dir, 'full_%d.pt' % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [8, 7])
def test_all_gather_descriptors_gloo_world2(tmp_path, n_total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(os.path.join(tmp_path, 'full_0.pt'))
    b = torch.load(os.path.join(tmp_path, 'full_1.pt'))
    assert torch.equal(a, b)


def test_shard_bounds_cover_batch_in_order():
    for n in (1, 7, 32, 256, 513):
        for world in (1, 2, 4, 8):
            edges = [shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    x = torch.randn(3, 5)
    assert all_gather_descriptors(x) is x


# ---------------------------------------------------------------- multi-staged step
class _ToyEncoder(torch.nn.Module):
    """Stands in for the encoder on CPU: {'x': (b, 6)} -> {'global': unit-norm (b, 5)}."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(6, 7)
        self.b = torch.nn.Linear(7, 5)

    def forward(self, batch):
        return {'global': torch.nn.functional.normalize(self.b(torch.tanh(self.a(batch['x']))), dim=1)}


def _toy_listwise_loss(emb, pos, neg):
    s = emb @ emb.t()
    loss = (torch.sigmoid((s * neg.float()).sum(1) - (s * pos.float()).sum(1))).mean()
    return loss, {'loss': loss.item()}


def _toy_data(n_total):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n_total, 6, generator=g)
    lab = torch.arange(n_total) // 2
    pos = (lab[:, None] == lab[None, :]) & ~torch.eye(n_total, dtype=torch.bool)
    neg = lab[:, None] != lab[None, :]
    return x, pos, neg


def _step_worker(rank, world, port, n_total, result_dir):
    from hotformerloc_amd.training import multistaged_training_step
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x, pos, neg = _toy_data(n_total)
        lo, hi = shard_bounds(n_total, rank, world)
        mine = x[lo:hi]
        half = (hi - lo + 1) // 2
        minibatches = [{'x': mine[:half]}, {'x': mine[half:]}]           # batch_split_size chunks
        model = _ToyEncoder()
        stats = multistaged_training_step(model, minibatches, pos, neg, _toy_listwise_loss, n_total=n_total)
        torch.save({'grads': [p.grad.clone() for p in model.parameters()], 'loss': stats['loss']},
                   os.path.join(result_dir, 'step_%d.pt' % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [8, 7])
def test_multistaged_step_gloo_world2_matches_single_process(tmp_path, n_total):
    """Two ranks, two minibatches each: the all-reduced parameter gradients equal those of one process that
    back-propagates the same listwise loss through the whole batch directly."""
    world = 2
    mp.spawn(_step_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'step_0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'step_1.pt'))
    x, pos, neg = _toy_data(n_total)
    model = _ToyEncoder()
    loss, _ = _toy_listwise_loss(model({'x': x})['global'], pos, neg)
    loss.backward()
    assert abs(r0['loss'] - loss.item()) < 1e-6 and abs(r1['loss'] - loss.item()) < 1e-6
    for g0, g1, p in zip(r0['grads'], r1['grads'], model.parameters()):
        assert torch.allclose(g0, g1, atol=0, rtol=0)
        assert torch.allclose(g0, p.grad, atol=1e-6), (g0 - p.grad).abs().max()


def test_multistaged_step_single_process_equals_direct():
    from hotformerloc_amd.training import multistaged_training_step
    x, pos, neg = _toy_data(10)
    model = _ToyEncoder()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    multistaged_training_step(model, [{'x': x[:4]}, {'x': x[4:7]}, {'x': x[7:]}], pos, neg, _toy_listwise_loss, opt)
    ref = _ToyEncoder()
    loss, _ = _toy_listwise_loss(ref({'x': x})['global'], pos, neg)
    loss.backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6)


class _ToyEncoderWithUnused(_ToyEncoder):
    def __init__(self):
        super().__init__()
        self.never_used = torch.nn.Parameter(torch.ones(3))           # no gradient on any rank
        self.sometimes = torch.nn.Parameter(torch.zeros(5))           # gradient on rank 0 only
        self.use_sometimes = False

    def forward(self, batch):
        y = self.b(torch.tanh(self.a(batch['x'])))
        if self.use_sometimes:
            y = y + self.sometimes
        return {'global': torch.nn.functional.normalize(y, dim=1)}


def _unused_worker(rank, world, port, result_dir):
    from hotformerloc_amd.training import multistaged_training_step
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x, pos, neg = _toy_data(8)
        lo, hi = shard_bounds(8, rank, world)
        model = _ToyEncoderWithUnused()
        model.use_sometimes = rank == 0
        multistaged_training_step(model, [{'x': x[lo:hi]}], pos, neg, _toy_listwise_loss, n_total=8)
        torch.save({'never': model.never_used.grad, 'sometimes': model.sometimes.grad},
                   os.path.join(result_dir, 'unused_%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_allreduce_keeps_grad_none_for_parameters_unused_on_every_rank(tmp_path):
    """A parameter no rank touched keeps grad None (AdamW then skips it, as the single-process reference step
    does); one that only some ranks touched is summed with zeros from the others."""
    mp.spawn(_unused_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'unused_0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'unused_1.pt'))
    assert r0['never'] is None and r1['never'] is None
    assert r0['sometimes'] is not None and torch.equal(r0['sometimes'], r1['sometimes'])
    assert r0['sometimes'].abs().sum() > 0


# ---------------------------------------------------------------- overlapped gradient reduction (round 3)
def _overlap_worker(rank, world, port, n_total, result_dir):
    from hotformerloc_amd.training import OverlappedGradReducer, multistaged_training_step
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x, pos, neg = _toy_data(n_total)
        lo, hi = shard_bounds(n_total, rank, world)
        mine = x[lo:hi]
        half = (hi - lo + 1) // 2
        minibatches = [{'x': mine[:half]}, {'x': mine[half:]}] if hi - lo > 1 else [{'x': mine}]
        model = _ToyEncoderWithUnused()
        model.use_sometimes = rank == 0
        reducer = OverlappedGradReducer(model.parameters(), bucket_bytes=64)     # tiny buckets: several per step
        out = []
        for step in range(4):
            if step == 2:
                model.use_sometimes = rank == world - 1          # the set of touched parameters moves to another rank
            if step == 3:
                model.use_sometimes = False                      # ... and then no rank touches it
            stats = multistaged_training_step(model, minibatches, pos, neg, _toy_listwise_loss, n_total=n_total,
                                              reducer=reducer)
            out.append({'grads': [None if p.grad is None else p.grad.clone() for p in model.parameters()],
                        'loss': stats['loss'], 'during_backward': reducer.launched_during_backward,
                        'buckets': len(reducer.buckets)})
        reducer.close()
        torch.save(out, os.path.join(result_dir, 'ov_%d.pt' % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_total', [(2, 8), (4, 10)])
def test_overlapped_gradient_reduction_matches_single_process(tmp_path, world, n_total):
    """OverlappedGradReducer behind the multi-staged step (gloo; world 4 with UNEVEN shards 3,3,2,2): step 0 learns which
    parameters receive gradients (plain reduction), steps 1-2 reduce bucket by bucket from the gradient hooks of the last
    backward -- same sums as one process back-propagating the whole batch, identical on every rank, `grad = None` kept for a
    parameter no rank uses, zeros contributed for one only some ranks use, and `grad = None` again in a step where a
    parameter of the learnt set receives no gradient on any rank (step 3; the reference's step leaves it None and AdamW skips it)."""
    mp.spawn(_overlap_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, 'ov_%d.pt' % r)) for r in range(world)]
    x, pos, neg = _toy_data(n_total)
    for step in range(4):
        # single-process gradients: `sometimes` is added on the rows of the rank that uses it
        ref = _ToyEncoderWithUnused()
        user = 0 if step < 2 else world - 1
        lo, hi = shard_bounds(n_total, user, world)
        y = ref.b(torch.tanh(ref.a(x)))
        if step < 3:
            y = torch.cat([y[:lo], y[lo:hi] + ref.sometimes, y[hi:]], 0)
        loss, _ = _toy_listwise_loss(torch.nn.functional.normalize(y, dim=1), pos, neg)
        loss.backward()
        want = [p.grad for p in ref.parameters()]
        for r in range(world):
            got = res[r][step]
            assert abs(got['loss'] - loss.item()) < 1e-6
            for g, w, (name, _) in zip(got['grads'], want, ref.named_parameters()):
                if name == 'never_used' or (name == 'sometimes' and step == 3):
                    assert g is None, (step, r, name)
                else:
                    assert g is not None and torch.allclose(g, w, atol=1e-6), (step, r, name)
            for g, g0 in zip(got['grads'], res[0][step]['grads']):
                assert (g is None and g0 is None) or torch.equal(g, g0)
        if step >= 1:                                   # overlapped steps really launched buckets from the hooks
            assert res[0][step]['buckets'] >= 2 and res[0][step]['during_backward'] >= 1


def _unarmed_worker(rank, world, port, result_dir):
    from hotformerloc_amd.training import OverlappedGradReducer
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x, _, _ = _toy_data(8)
        lo, hi = shard_bounds(8, rank, world)
        model = _ToyEncoder()
        reducer = OverlappedGradReducer(model.parameters(), bucket_bytes=64)
        out = []
        for step in range(3):
            model.zero_grad(set_to_none=True)
            y = model({'x': x[lo:hi]})['global'].sum()
            if step >= 1 and rank == 0:
                reducer.arm()                          # rank 1 never arms: it must still issue the same bucket sequence
            y.backward()
            reducer.finish()
            out.append([p.grad.clone() for p in model.parameters()])
        if rank == 0:                                  # an exception in the backward: no stale state in the next step
            reducer.arm()
            reducer.reset()
            assert not reducer.armed and reducer._works == []
        reducer.close()
        torch.save(out, os.path.join(result_dir, 'ua_%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_overlapped_reducer_same_collectives_when_a_rank_never_arms(tmp_path):
    """A rank that does not call arm() in a step (no local minibatch) sends every bucket from finish(), in the same order
    as the ranks that launched from their hooks: no mismatched collectives, same sums."""
    mp.spawn(_unarmed_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'ua_0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'ua_1.pt'))
    x, _, _ = _toy_data(8)
    ref = _ToyEncoder()
    ref({'x': x})['global'].sum().backward()
    for step in range(3):
        for g0, g1, p in zip(r0[step], r1[step], ref.parameters()):
            assert torch.equal(g0, g1) and torch.allclose(g0, p.grad, atol=1e-6)


def test_bench_rank_slices_concatenate_to_the_global_batch():
    """bench.py gives rank r the clouds [r B, (r + 1) B) of ONE global synthetic batch (weak scaling): the
    per-rank slices concatenated equal the single-process batch of N B clouds, in order, for both workloads."""
    import importlib.util
    import types
    import numpy as np
    from hotformerloc_amd import synthetic as syn
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(__file__)), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    params = types.SimpleNamespace(coordinates='cylindrical')
    for pmax in (None, 2048):
        args = types.SimpleNamespace(batch=3, points=512, points_max=pmax)
        world = 2
        parts = [bench.bench_clouds(syn, params, args, r) for r in range(world)]
        whole = bench.bench_clouds(syn, params, types.SimpleNamespace(batch=6, points=512, points_max=pmax), 0)
        flat = [c for part in parts for c in part]
        assert len(flat) == len(whole) == 6
        for a, b in zip(flat, whole):
            assert a.shape == b.shape and np.array_equal(a, b)


# ---------------------------------------------------------------- bench.py launcher + per-rank reporting (round 4)
_BENCH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')


def _bench_env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


def test_bench_self_launch_command():
    """`python bench.py --gpus N` with no RANK in the environment starts its own ranks: the child command is the driver's
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    ...`), the user's flags are passed through, and --gpus 1 launches nothing."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, _BENCH, '--gpus', '4', '--steps', '3', '--warmup', '1', '--dry-launch'],
                       capture_output=True, text=True, env=_bench_env(), timeout=120)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])['launch']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert int(cmd[cmd.index('--master-port') + 1]) > 0
    tail = cmd[cmd.index(_BENCH) + 1:]
    assert tail == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    r = subprocess.run([sys.executable, _BENCH, '--dry-launch'], capture_output=True, text=True, env=_bench_env(), timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])['launch'] is None


def test_bench_self_launch_reports_missing_gpus_from_the_children():
    """On a box with fewer GPUs than ranks the CHILDREN say so ("need N GPUs") and the parent relays their non-zero exit
    code -- no usage error from the parent, no hang."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box has the GPUs')
    r = subprocess.run([sys.executable, _BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, env=_bench_env(), timeout=600)
    assert r.returncode != 0
    assert 'need 2 GPUs on this node' in r.stderr, r.stderr[-800:]


def _rank_report_worker(rank, world, port, result_dir):
    import importlib.util
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        spec = importlib.util.spec_from_file_location('bench_mod', _BENCH)
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        from hotformerloc_amd.distributed import all_gather_descriptors
        # stub step: this rank's "descriptors" + the bench's collective, then the bench's own reporting code
        local = torch.full((3, 4), float(rank))
        g = all_gather_descriptors(local, 3 * world, force=True)
        per_rank, worst = bench.rank_report(dist, 0.010 * (rank + 1), 5, torch.device('cpu'))
        torch.save({'per_rank': per_rank, 'worst': worst, 'gathered': g}, os.path.join(result_dir, 'rr_%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_bench_rank_report_is_max_over_ranks(tmp_path):
    """bench.py's N > 1 reporting (per-rank ms/step in rank order, MAX over ranks as the timed region) under gloo, world 2,
    with a stub step that runs the bench's descriptor all-gather."""
    mp.spawn(_rank_report_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(os.path.join(tmp_path, 'rr_%d.pt' % r))
        assert got['per_rank'] == [2.0, 4.0]
        assert abs(got['worst'] - 0.020) < 1e-12
        assert torch.equal(got['gathered'], torch.tensor([0.0, 0, 0, 1, 1, 1])[:, None].expand(6, 4))
