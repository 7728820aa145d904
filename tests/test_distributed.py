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
