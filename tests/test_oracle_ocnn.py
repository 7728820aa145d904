"""Pin `oracle.ocnn_ref` against ocnn's own golden vectors (the data files of the
reference's only test suite, `libs/dwconv/test/data/`, copied to tests/golden/ocnn/;
loader semantics follow `libs/dwconv/test/utils.py:15-39`)."""

import os

import numpy as np
import pytest
import torch

from oracle.ocnn_ref import Octree, Points, merge_octrees, key2xyz, xyz2key


def _load(golden_dir, i):
    d = np.load(os.path.join(golden_dir, 'ocnn', 'test_%03d.npz' % i))
    pc = Points(torch.from_numpy(d['points']), torch.from_numpy(d['normals']))
    o = Octree(int(d['depth']), int(d['full_depth']))
    o.build_octree(pc)
    return o, d


@pytest.mark.parametrize('i', [1, 2, 3, 4, 5])
def test_build_octree_bit_exact(golden_dir, i):
    o, d = _load(golden_dir, i)
    assert np.array_equal(o.nnum.numpy(), d['nnum'])
    assert np.array_equal(o.nnum_nempty.numpy(), d['nnum_nempty'])
    assert np.array_equal(torch.cat(o.keys).numpy(), d['key'])
    assert np.array_equal(torch.cat(o.children).numpy(), d['child'])
    f = o.get_input_feature('ND', nempty=True).numpy()
    assert np.abs(f - d['feature']).max() < 2e-6


def test_merge_and_neigh_bit_exact(golden_dir):
    o4, _ = _load(golden_dir, 4)
    o5, _ = _load(golden_dir, 5)
    b = np.load(os.path.join(golden_dir, 'ocnn', 'batch_45.npz'))
    m = merge_octrees([o4, o5])
    m.construct_all_neigh()
    assert np.array_equal(torch.cat(m.keys).numpy(), b['key'])
    assert np.array_equal(torch.cat(m.children).numpy(), b['child'])
    assert np.array_equal(m.nnum.numpy(), b['nnum'])
    assert np.array_equal(m.nnum_nempty.numpy(), b['nnum_nempty'])
    neigh = torch.cat(m.neighs[1:]).numpy()
    assert neigh.shape == b['neigh'].shape
    assert np.array_equal(neigh, b['neigh'])
    f = m.get_input_feature('ND', nempty=True).numpy()
    assert np.abs(f - b['feature']).max() < 2e-6
    assert m.batch_size == 2 and tuple(m.batch_nnum_nempty.shape) == (7, 2)


def test_get_neigh_variants(golden_dir):
    o4, _ = _load(golden_dir, 4)
    o5, _ = _load(golden_dir, 5)
    m = merge_octrees([o4, o5])
    m.construct_all_neigh()
    d = 5
    ne = m.get_neigh(d, '333', 1, nempty=True)
    assert ne.shape == (int(m.nnum_nempty[d]), 27)
    assert torch.equal(ne[:, 13], torch.arange(ne.shape[0]))            # column 13 = self
    s2 = m.get_neigh(d, '222', 2, nempty=True)
    assert s2.shape == (int(m.nnum_nempty[d - 1]), 8)
    # stride-2 '222' rows are the eight children of each non-empty parent
    assert torch.equal(s2, m.children[d].view(-1, 8).long())
    # the inverse table used for the data gradient is consistent (dwconv.cu:74-85)
    valid = ne >= 0
    h = torch.arange(ne.shape[0]).unsqueeze(1).expand_as(ne)
    inv = torch.full_like(ne, -1)
    k = torch.arange(27).unsqueeze(0).expand_as(ne)
    inv[ne[valid], k[valid]] = h[valid]
    assert torch.equal(inv, ne[:, torch.arange(26, -1, -1)])            # symmetric stencil


def test_key_roundtrip():
    g = torch.Generator().manual_seed(0)
    for depth in (1, 5, 7, 9, 12):
        xyz = torch.randint(0, 1 << depth, (1000, 3), generator=g)
        b = torch.randint(0, 300, (1000,), generator=g)
        key = xyz2key(xyz[:, 0], xyz[:, 1], xyz[:, 2], b, depth)
        x, y, z, bb = key2xyz(key, depth)
        assert torch.equal(torch.stack([x, y, z], 1), xyz) and torch.equal(bb, b)
    # x is the most significant bit of each triple
    assert int(xyz2key(torch.tensor([1]), torch.tensor([0]), torch.tensor([0]), None, 1)) == 4
