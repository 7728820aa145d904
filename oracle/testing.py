"""Helpers shared by tests/, smoke() and bench.py's cpu_baseline leg.
TEST INFRASTRUCTURE (see oracle/__init__.py)."""

import os
from typing import Dict, List

import numpy as np
import torch

from hotformerloc_amd import synthetic as syn
from oracle.ocnn_ref import Octree, Points, merge_octrees

_NAME = {'wild-places': 'wild-places', 'cs-wild-places': 'cs-wild-places', 'oxford': 'oxford',
         'cs-campus3d': 'cs-campus3d'}


def load_case(golden_dir: str, case: str) -> dict:
    z = np.load(os.path.join(golden_dir, 'model_%s.npz' % case))
    g = {k: z[k] for k in z.files}
    g['cfg'] = _NAME[str(g['cfg'])]
    g['octree_depth'] = int(g['octree_depth'])
    if 'workload' in g:
        # full-size BASELINE workloads (oracle/gen_golden.py::WORKLOAD_CASES): the fixture holds the arguments of the product's
        # own cloud generator (bench.py::bench_clouds), not the points; regenerate and check them against the stored checksum
        from hotformerloc_amd import synthetic as syn
        from hotformerloc_amd import load_config
        cid, batch, n_points, n_points_max = (int(v) for v in g['workload'])
        coords = load_config(g['cfg'])[0].coordinates
        if n_points_max:
            clouds = []
            for i in range(batch):
                clouds += syn.make_clouds(cid, 1, n_points, coords, kind='forest' if i % 2 == 0 else 'ball',
                                          n_points_max=n_points_max, first_index=i)
        else:
            clouds = syn.make_clouds(cid, batch, n_points, coords)
        assert [c.shape[0] for c in clouds] == g['n_points'].tolist()
        allp = np.concatenate(clouds, 0).astype(np.float64)
        assert np.allclose([allp.sum(), (allp ** 2).sum()], g['points_sum'], rtol=1e-12, atol=1e-9), 'generator drifted'
        g['clouds'] = clouds
        g['profile'] = str(g['profile'])
        return g
    offs = np.concatenate([[0], np.cumsum(g['n_points'])])
    g['clouds'] = [g['points'][offs[i]:offs[i + 1]] for i in range(len(g['n_points']))]
    return g


def oracle_octree(clouds: List[np.ndarray], depth: int, full_depth: int = 2):
    """`create_batch` (datasets/dataset_utils.py:74-98) + `construct_all_neigh`
    (misc/torch_utils.py:47-51) with the restated ocnn."""
    octs = []
    for pc in clouds:
        o = Octree(depth, full_depth)
        o.build_octree(Points(torch.from_numpy(np.ascontiguousarray(pc, dtype=np.float32))))
        octs.append(o)
    m = merge_octrees(octs)
    m.construct_all_neigh()
    return m


def state_dict_spec(params) -> Dict[str, tuple]:
    """Names and shapes of the reference state_dict, derived
    from the product's own module tree (tests assert it equals the reference's)."""
    from hotformerloc_amd.model_factory import model_factory
    model = model_factory(params)
    return {k: tuple(v.shape) for k, v in model.state_dict().items()}


def synthetic_state_dict(params, profile: str = 'stress') -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(syn.synthetic_tensor(k, s, profile))
            for k, s in state_dict_spec(params).items()}
