"""alternative pooling heads and ablation switches, against goldens produced by the reference's own
`model_factory` + forward for each variant cfg (`oracle/gen_golden_variants.py`).  The state_dict layout check runs
on CPU; descriptor parity (<= 1e-3 relative, as for the shipped configs) needs the GPU."""

import json
import os
import re

import numpy as np
import pytest
import torch

from hotformerloc_amd import build_batch_octree, model_factory
from hotformerloc_amd import synthetic as syn
from hotformerloc_amd.params import CONFIG_DIR, ModelParams
from oracle.gen_golden_variants import DEPTH, VARIANTS, variant_clouds

REL_TOL = 1e-3


def _params(name, tmp_path):
    base, over, _, _ = VARIANTS[name]
    src = open(os.path.join(CONFIG_DIR, base.replace('-', '_') + '.ini')).read()
    for k, v in over.items():
        pat = re.compile(r'^%s\s*=.*$' % re.escape(k), flags=re.M)
        line = '%s = %s' % (k, v)
        src = pat.sub(line, src) if pat.search(src) else src.rstrip('\n') + '\n' + line + '\n'
    path = tmp_path / ('%s.ini' % name)
    path.write_text(src)
    return ModelParams(str(path))


@pytest.mark.parametrize('name', sorted(VARIANTS))
def test_variant_state_dict_matches_reference_layout(golden_dir, tmp_path, name):
    model = model_factory(_params(name, tmp_path))
    spec = json.load(open(os.path.join(golden_dir, 'state_dict_variant_%s.json' % name)))
    mine = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    assert mine == spec


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(VARIANTS))
def test_variant_descriptors_match_reference_golden(golden_dir, tmp_path, name):
    params = _params(name, tmp_path)
    base, _, spec, seed = VARIANTS[name]
    g = np.load(os.path.join(golden_dir, 'variant_%s.npz' % name))
    model = model_factory(params)
    syn.fill_synthetic_weights(model, 'stress')
    model = model.cuda().eval()
    clouds = variant_clouds(spec, seed, params.coordinates)
    octree = build_batch_octree(clouds, DEPTH[base], 2, 'cuda')
    assert np.array_equal(octree.nnum_nempty.numpy(), g['nnum_nempty'])
    with torch.inference_mode():
        y = model({'octree': octree})['global'].cpu().numpy()
    want = g['descriptors']
    assert y.shape == want.shape and np.isfinite(y).all()
    rel = np.linalg.norm(y.astype(np.float64) - want, axis=1) / np.linalg.norm(want, axis=1)
    print(name, 'descriptor rel-L2', rel.tolist())
    assert rel.max() <= REL_TOL, rel


def test_ct_size_other_than_one_is_rejected_like_the_reference(tmp_path):
    """`ct_size != 1` (relay tokens per window) is the one ModelParams option this package refuses: the reference's own model
    cannot run it either -- `RelayTokenInitialiser.forward` views the windows as (-1, K // G, C) against a (windows, K) mask
    and raises (models/hotformerloc_backbone.py:354-357, marked "TODO: Make this work with rt_size > 1").  When the
    reference tree is present the failure is reproduced live."""
    src = open(os.path.join(CONFIG_DIR, 'wild_places.ini')).read()
    src = re.sub(r'^ct_size\s*=.*$', 'ct_size = 2', src, flags=re.M) if re.search(r'^ct_size\s*=', src, flags=re.M) \
        else src.rstrip('\n') + '\nct_size = 2\n'
    path = tmp_path / 'ct2.ini'
    path.write_text(src)
    with pytest.raises(NotImplementedError, match='ct_size'):
        model_factory(ModelParams(str(path)))
    from oracle import ref_import
    if not ref_import.reference_available():
        return
    from oracle.gen_golden_variants import variant_cfg_text
    from oracle.ocnn_ref import Octree, Points, merge_octrees
    ref_cfg = tmp_path / 'ref_ct2.txt'
    ref_cfg.write_text(variant_cfg_text('wild-places', {'ct_size': '2'}, os.path.join(ref_import.REFERENCE_ROOT, 'models'),
                                        '.txt'))
    model, params = ref_import.reference_model(str(ref_cfg))
    octs = []
    for pc in variant_clouds([(600, 'ball'), (500, 'forest')], 5, params.coordinates):
        o = Octree(7, 2)
        o.build_octree(Points(torch.from_numpy(pc)))
        octs.append(o)
    octree = merge_octrees(octs)
    octree.construct_all_neigh()
    with pytest.raises(RuntimeError, match='must match the size'), torch.no_grad():
        model({'octree': octree})
