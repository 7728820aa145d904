"""Build-container only: goldens of the reference model with NON-default heads / ablation switches
-> tests/golden/variant_<name>.npz + tests/golden/state_dict_variant_<name>.json.

Each variant is one of the reference's own model cfg files with a few keys changed (written to a temp file), built by
the reference's `model_factory`, filled with the closed-form synthetic weights and run on CPU by the reference's own
forward over `oracle.ocnn_ref` (see oracle/ref_import.py).  Inputs are regenerated from the seeds by the tests.

Usage:  python -m oracle.gen_golden_variants [name ...]"""
import json
import os
import re
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import                                   # noqa: E402
from oracle.ocnn_ref import Octree, Points, merge_octrees        # noqa: E402
from hotformerloc_amd import synthetic as syn                    # noqa: E402

# name -> (base cfg, {key: value} overrides, [(n_points, kind)], seed)
VARIANTS = {
    'octgem':          ('wild-places', {'pooling': 'OctGeM'}, [(3000, 'ball'), (2500, 'forest')], 61),
    'pyr_octgem':      ('cs-wild-places', {'pooling': 'PyramidOctGeM'}, [(4096, 'forest'), (3000, 'ball')], 62),
    'pyr_octgem_gc':   ('wild-places', {'pooling': 'PyramidOctGeMgc'}, [(3000, 'ball'), (2000, 'ball'), (1500, 'forest')], 63),
    'attnpool_mixer':  ('wild-places', {'pooling': 'AttnPoolMixer', 'k_pooled_tokens': '64'}, [(3000, 'ball'), (2500, 'forest')], 64),
    'attnpool_gem':    ('cs-wild-places', {'pooling': 'AttnPoolGeM', 'k_pooled_tokens': '32'}, [(4096, 'forest'), (3000, 'ball')], 65),
    'disable_rt':      ('wild-places', {'disable_rt': 'True'}, [(3000, 'ball'), (2500, 'forest')], 66),
    'xcpe':            ('cs-wild-places', {'xCPE': 'True'}, [(3000, 'forest'), (2000, 'ball')], 67),
    'layer_scale':     ('wild-places', {'layer_scale': '0.1'}, [(3000, 'ball'), (2500, 'forest')], 68),
    # round 3: the rest of SURVEY 8 f4
    'ct_prop':         ('wild-places', {'ct_propagation': 'True'}, [(3000, 'ball'), (2500, 'forest')], 69),
    'ct_prop_scale':   ('cs-wild-places', {'ct_propagation': 'True', 'ct_propagation_scale': '0.5'},
                        [(4096, 'forest'), (3000, 'ball')], 70),
    'level_channels':  ('wild-places', {'channels': '128,256,192,128', 'num_heads': '8,16,12,8', 'feature_size': '256'},
                        [(3000, 'ball'), (2500, 'forest'), (1800, 'ball')], 71),
    'level_channels_gem': ('wild-places', {'channels': '128,256,192,128', 'num_heads': '8,16,12,8', 'feature_size': '256',
                                           'pooling': 'PyramidOctGeM'}, [(3000, 'ball'), (2500, 'forest')], 72),
    'no_input_downsample': ('wild-places', {'downsample_input_embeddings': 'False'}, [(1500, 'ball'), (1200, 'forest')], 73),
}
DEPTH = {'wild-places': 7, 'cs-wild-places': 7}


def variant_cfg_text(base: str, overrides: dict, src_dir: str, suffix: str) -> str:
    text = open(os.path.join(src_dir, 'hotformerloc_%s_cfg%s' % (base, suffix))).read()
    for k, v in overrides.items():
        pat = re.compile(r'^%s\s*=.*$' % re.escape(k), flags=re.M)
        line = '%s = %s' % (k, v)
        text = pat.sub(line, text) if pat.search(text) else text.rstrip('\n') + '\n' + line + '\n'
    return text


def variant_clouds(spec, seed, coordinates):
    out = []
    for i, (n, kind) in enumerate(spec):
        pc = syn.unit_ball_cloud(1000 * seed + i, n) if kind == 'ball' else syn.forest_cloud(1000 * seed + i, n)
        out.append(syn.cylindrical(pc) if coordinates == 'cylindrical' else pc)
    return out


def main():
    dst = os.path.join(ROOT, 'tests', 'golden')
    torch.set_num_threads(os.cpu_count())
    for name in (sys.argv[1:] or VARIANTS):
        base, over, spec, seed = VARIANTS[name]
        text = variant_cfg_text(base, over, os.path.join(ref_import.REFERENCE_ROOT, 'models'), '.txt')
        with tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False) as f:
            f.write(text)
        model, params = ref_import.reference_model(f.name)
        os.unlink(f.name)
        syn.fill_synthetic_weights(model, 'stress')
        clouds = variant_clouds(spec, seed, params.coordinates)
        octs = []
        for pc in clouds:
            o = Octree(DEPTH[base], 2)
            o.build_octree(Points(torch.from_numpy(pc)))
            octs.append(o)
        octree = merge_octrees(octs)
        octree.construct_all_neigh()
        with torch.no_grad():
            y = model({'octree': octree})['global'].numpy()
        assert np.isfinite(y).all(), name
        np.savez_compressed(os.path.join(dst, 'variant_%s.npz' % name), descriptors=y,
                            nnum_nempty=octree.nnum_nempty.numpy())
        spec_sd = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        with open(os.path.join(dst, 'state_dict_variant_%s.json' % name), 'w') as fj:
            json.dump(spec_sd, fj)
        print(name, y.shape, 'norms', np.round(np.linalg.norm(y, axis=1), 4).tolist(), len(spec_sd), 'tensors',
              sum(p.numel() for p in model.parameters()), 'parameters')


if __name__ == '__main__':
    main()
