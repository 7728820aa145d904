"""`hotformerloc_amd.preprocess.prepare_clouds` (HIP, one launch per batch) against goldens produced by the
reference's own `Normalize` + mask + `CylindricalCoordinates` sequence (`eval/pnv_evaluate.py:158-171`,
`oracle/gen_golden_coords.py`).  Normalisation, masks and the host-side transform: bit-exact.  The all-device
transform: equal up to the 1-2 ulp of atan2f, measured here."""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hotformerloc_amd import build_batch_octree
from hotformerloc_amd.preprocess import prepare_clouds
from oracle.gen_golden_coords import CASES, raw_cloud


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'preprocess.npz'))


def test_prepare_clouds_bit_exact_per_case(golden_dir):
    g = _golden(golden_dir)
    for name, (seed, n, kind, extent, offset, normalize, coords) in CASES.items():
        raw = raw_cloud(seed, n, kind, extent, offset)
        got = prepare_clouds([raw], coordinates=coords, normalize=normalize)[0]
        assert got.is_cuda and got.dtype == torch.float32
        assert np.array_equal(got.cpu().numpy(), g[name + '_out']), name
        # masks + normalisation alone (cartesian call on the same cloud keeps the |x| <= 1 mask only)
        if coords == 'cartesian':
            assert np.array_equal(got.cpu().numpy(), g[name + '_masked']), name


def test_prepare_clouds_batched_and_feeds_the_octree_build(golden_dir):
    g = _golden(golden_dir)
    names = ['wp_forest', 'wp_ball', 'tiny']                        # all cylindrical + normalised: one batch
    raws = [raw_cloud(*CASES[k][:5]) for k in names]
    got = prepare_clouds(raws, coordinates='cylindrical', normalize=True)
    for k, t in zip(names, got):
        assert np.array_equal(t.cpu().numpy(), g[k + '_out']), k
    a = build_batch_octree(got, 7, 2, 'cuda')
    b = build_batch_octree([g[k + '_out'] for k in names], 7, 2, 'cuda')
    assert torch.equal(a.nnum_nempty, b.nnum_nempty)
    for d in range(8):
        assert torch.equal(a.nkeys[d], b.nkeys[d])


def test_device_side_cylindrical_transform_within_ulps(golden_dir):
    """cylindrical='device': same masks, rho / z bit-exact, phi within 2 ulp of the reference's CPU atan2 path; the
    fraction of points whose depth-7 cell changes is reported (expected ~1e-5)."""
    g = _golden(golden_dir)
    moved = total = 0
    for name in ('wp_forest', 'wp_ball', 'boundary_cyl'):
        seed, n, kind, extent, offset, normalize, coords = CASES[name]
        raw = raw_cloud(seed, n, kind, extent, offset)
        got = prepare_clouds([raw], coordinates='cylindrical', normalize=normalize, cylindrical='device')[0].cpu().numpy()
        want = g[name + '_out']
        assert got.shape == want.shape, name
        assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 2], want[:, 2]), name
        # phi wraps at +-pi (atan2(+-0, -x)): compare on the circle
        dphi = np.abs(got[:, 1].astype(np.float64) - want[:, 1])
        dphi = np.minimum(dphi, 2.0 - dphi)
        assert dphi.max() <= 3 * 2.0 ** -24, (name, dphi.max())
        cell = lambda a: np.clip(np.floor((a.astype(np.float64) + 1.0) * 64.0), 0, 127)
        moved += int((cell(got) != cell(want)).any(axis=1).sum())
        total += len(want)
    print('device-side cylindrical: %d of %d points changed depth-7 cell' % (moved, total))
    assert moved <= max(3, total // 1000)


def test_prepare_clouds_rejects_unsupported_modes():
    raw = raw_cloud(*CASES['tiny'][:5])
    with pytest.raises(NotImplementedError):
        prepare_clouds([raw], unit_sphere_norm=True)
    with pytest.raises(NotImplementedError):
        prepare_clouds([raw], scale_factor=30.0)
