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
    if 'profile' in g:                       # full-size BASELINE workloads (oracle/gen_golden.py::WORKLOAD_CASES)
        g['profile'] = str(g['profile'])
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
