"""Build-container only: golden vectors of the reference's raw-cloud pre-steps -> tests/golden/preprocess.npz.

Executes the reference's OWN `Normalize` (`/root/reference/datasets/augmentation.py:185-236`) and
`CylindricalCoordinates` (`/root/reference/datasets/coordinate_utils.py:68-116`) in the sequence of
`/root/reference/eval/pnv_evaluate.py:158-171` on closed-form raw clouds (metres, not normalised), including points
engineered to sit on the |x| = 1 and |xy| = 1 mask boundaries after normalisation.  `augmentation.py` imports
torchvision (absent from this image) for an unrelated `transforms.Compose`; an empty module object stands in for it at
generation time only.  Inputs are regenerated from the seeds by the tests; the file pins outputs.

Usage:  python -m oracle.gen_golden_coords"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hotformerloc_amd import synthetic as syn    # noqa: E402

REF = os.environ.get('HOTFORMERLOC_REFERENCE', '/root/reference')

# case -> (seed, n, kind, extent in metres (x, y, z), offset in metres, normalize, coordinates)
CASES = {
    'wp_forest':    (501, 6000, 'forest', (60.0, 60.0, 30.0), (250.0, -120.0, 12.0), True, 'cylindrical'),
    'wp_ball':      (502, 4096, 'ball', (35.0, 20.0, 8.0), (0.0, 0.0, 0.0), True, 'cylindrical'),
    'cs_forest':    (503, 9000, 'forest', (45.0, 45.0, 25.0), (-3.0, 7.0, 1.0), True, 'cartesian'),
    'oxford_ready': (504, 4096, 'ball', (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), False, 'cartesian'),
    'boundary_cyl': (505, 2048, 'boundary', (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), False, 'cylindrical'),
    'tiny':         (506, 3, 'ball', (5.0, 9.0, 2.0), (1.0, 2.0, 3.0), True, 'cylindrical'),
}


def raw_cloud(seed, n, kind, extent, offset):
    """Closed-form raw cloud (float32, metres)."""
    if kind == 'boundary':
        # already-normalised cloud hugging both mask boundaries: |x| = 1 +- a few ulp and |xy| = 1 +- a few ulp
        u = syn.hash_uniform(seed, n * 4).reshape(n, 4)
        ang = (u[:, 0] * np.pi).astype(np.float64)
        r = 1.0 + np.round(u[:, 1] * 6) * 2.0 ** -24                     # radius within +-6 ulp of 1
        pts = np.stack([r * np.cos(ang), r * np.sin(ang), u[:, 2] * 0.9], 1)
        k = n // 4                                                       # a quarter: coordinates at +-1 +- ulps
        pts[:k, 0] = np.sign(u[:k, 3]) * (1.0 + np.round(u[:k, 1] * 3) * 2.0 ** -24)
        pts[:k, 1] = u[:k, 2] * 0.3
        pts[k:2 * k, 2] = np.sign(u[k:2 * k, 3]) * (1.0 + np.round(u[k:2 * k, 1] * 3) * 2.0 ** -24)
        pts[0] = [1.0, 0.0, 0.0]
        pts[1] = [-1.0, 0.0, 1.0]                                        # phi = pi exactly
        pts[2] = [0.0, 0.0, 0.0]                                         # atan2(0, 0)
        pts[3] = [0.0, -1.0, -1.0]
        pts[4] = [-1.0, -0.0, 0.5]                                       # phi = -pi (negative zero y)
        return pts.astype(np.float32)
    base = syn.forest_cloud(seed, n) if kind == 'forest' else syn.unit_ball_cloud(seed, n)
    return (base.astype(np.float64) * np.asarray(extent) + np.asarray(offset)).astype(np.float32)


def _reference_classes():
    for name in ('torchvision', 'torchvision.transforms'):
        sys.modules.setdefault(name, types.ModuleType(name))            # generation-time stand-in (see docstring)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    spec = importlib.util.spec_from_file_location('ref_augmentation', os.path.join(REF, 'datasets', 'augmentation.py'))
    aug = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(aug)
    spec = importlib.util.spec_from_file_location('ref_coordinate_utils',
                                                  os.path.join(REF, 'datasets', 'coordinate_utils.py'))
    cu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cu)
    return aug.Normalize, cu.CylindricalCoordinates


def reference_sequence(data, normalize, coordinates, Normalize, Cyl):
    """eval/pnv_evaluate.py:158-171 with the reference's own callables."""
    stages = {}
    if normalize:
        data = Normalize(scale_factor=None, unit_sphere_norm=False)(data)
    stages['normalized'] = data.clone()
    mask = torch.all(abs(data) <= 1.0, dim=1)
    data = data[mask]
    if coordinates == 'cylindrical':
        data_norm = torch.linalg.norm(data[:, :2], dim=1)[:, None]
        mask = torch.all(data_norm <= 1.0, dim=1)
        data = data[mask]
        stages['masked'] = data.clone()
        data = Cyl(use_octree=True)(data)
    else:
        stages['masked'] = data.clone()
    return data, stages


def main():
    Normalize, Cyl = _reference_classes()
    out = {}
    for name, (seed, n, kind, extent, offset, normalize, coords) in CASES.items():
        raw = torch.from_numpy(raw_cloud(seed, n, kind, extent, offset))
        final, stages = reference_sequence(raw.clone(), normalize, coords, Normalize, Cyl)
        out[name + '_out'] = final.numpy()
        out[name + '_normalized'] = stages['normalized'].numpy()
        out[name + '_masked'] = stages['masked'].numpy()
        print(name, 'raw', tuple(raw.shape), '-> kept', tuple(final.shape))
    # a3 alone: the reference transform on the product's synthetic bench clouds before the transform
    for i in range(2):
        pc = torch.from_numpy(syn.unit_ball_cloud(2000 + i, 4096))
        out['a3_ball_%d' % i] = Cyl(use_octree=True)(pc.clone()).numpy()
    path = os.path.join(ROOT, 'tests', 'golden', 'preprocess.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
