"""Pin `oracle.hotformer_ref` against the reference's own model outputs.

The fixtures tests/golden/model_*.npz were produced by `oracle/gen_golden.py`, which
imports the reference's Python model files in the build container; here only the
oracle runs (no reference needed), so this test also runs on the GPU box."""

import os

import numpy as np
import pytest
import torch

from hotformerloc_amd.params import load_config
from hotformerloc_amd import synthetic as syn
from oracle import hotformer_ref
from oracle.ocnn_ref import Octree, Points, merge_octrees
from oracle.testing import load_case, oracle_octree, synthetic_state_dict

CASES = ['wild_places_b1', 'wild_places_ragged', 'cs_wild_places_b2', 'oxford_b2', 'wild_places_b3', 'cs_campus3d_b2']


@pytest.mark.parametrize('case', CASES)
def test_oracle_matches_reference_golden(golden_dir, case):
    g = load_case(golden_dir, case)
    params, depth = load_config(g['cfg'])
    assert depth == g['octree_depth']
    octree = oracle_octree(g['clouds'], depth)
    assert np.array_equal(octree.nnum_nempty.numpy(), g['nnum_nempty'])
    sd = synthetic_state_dict(params)
    cap = {}
    y = hotformer_ref.forward(sd, params, octree, cap).numpy()
    ref = g['descriptors']
    rel = np.linalg.norm(y - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 2e-5, rel
    for name, val in (('patch_embed', cap['patch_embed']), ('octf_out', cap['octf_out'])):
        head = g[name + '_head']
        assert np.abs(val[:head.shape[0]].numpy() - head).max() < 1e-4
        s = val.double()
        assert abs(s.sum().item() - g[name + '_sum'][0]) < 1e-3 * max(1.0, abs(g[name + '_sum'][0]))
    for d in cap['plan'].pyramid_depths:
        for kind in ('feat_final', 'rt_final'):
            head = g['%s_%d_head' % (kind, d)]
            val = cap['%s.%d' % (kind, d)]
            assert np.abs(val[:head.shape[0]].numpy() - head).max() < 2e-4


def test_oracle_matches_reference_on_the_bench_workload(golden_dir):
    """The batch `bench.py` times (32 clouds x 4096 points, Wild-Places cfg, 'init' weights): the oracle -- which the bench's
    `parity` block and `cpu_baseline` run -- against the reference's own descriptors for that batch
    (oracle/gen_golden.py::WORKLOAD_CASES)."""
    g = load_case(golden_dir, 'wild_places_b32')
    params, depth = load_config(g['cfg'])
    octree = oracle_octree(g['clouds'], depth)
    assert np.array_equal(octree.nnum_nempty.numpy(), g['nnum_nempty'])
    y = hotformer_ref.forward(synthetic_state_dict(params, g['profile']), params, octree).numpy()
    ref = g['descriptors']
    rel = np.linalg.norm(y - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 2e-5, rel


def test_oracle_matches_reference_live():
    """Only where /root/reference exists (build container): rerun the reference."""
    from oracle import ref_import
    if not ref_import.reference_available():
        pytest.skip('reference tree not present (GPU box)')
    model, params = ref_import.reference_model(
        os.path.join(ref_import.REFERENCE_ROOT, 'models', 'hotformerloc_cs-wild-places_cfg.txt'))
    syn.fill_synthetic_weights(model, 'stress')
    clouds = [syn.forest_cloud(77, 1500), syn.unit_ball_cloud(78, 900)]
    octree = oracle_octree(clouds, 7)
    with torch.inference_mode():
        ref = model({'octree': octree})['global'].numpy()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    y = hotformer_ref.forward(sd, params, octree).numpy()
    rel = np.linalg.norm(y - ref, axis=1) / np.linalg.norm(ref, axis=1)
    assert rel.max() < 2e-5
