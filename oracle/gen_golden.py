"""Generate the golden model fixtures under tests/golden/ (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Imports the reference's own Python
model files from /root/reference over `oracle.ocnn_ref` (see oracle/ref_import.py),
fills them with the closed-form synthetic weights of `hotformerloc_amd.synthetic`
(profile 'stress'), runs the REFERENCE forward on CPU and stores inputs + outputs:

    tests/golden/model_<case>.npz
        cfg            name of the model cfg (file committed under hotformerloc_amd/configs/)
        octree_depth   int
        n_points       (B,) points per cloud
        points         (sum n, 3) float32 -- clouds *after* the coordinate transform,
                       i.e. exactly what `Points(...)` receives (dataset_utils.py:85-90)
        descriptors    (B, 256) float32  -- reference `model(batch)['global']`
        patch_embed_head / octf_out_head / feat_final_<d>_head / rt_final_<d>_head
                       first 32 rows of reference intermediates (forward hooks)
        *_sum          float64 checksums (sum, sum of squares) of the full intermediates

Usage:  python -m oracle.gen_golden            (writes all cases)
"""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import                                   # noqa: E402
from oracle.ocnn_ref import Octree, Points, merge_octrees        # noqa: E402
from hotformerloc_amd import synthetic as syn                    # noqa: E402

# case -> (cfg, octree depth, config_id, list of (n_points, kind))
CASES = {
    'wild_places_b3':     ('wild-places', 7, 1, [(4096, 'ball')] * 3),
    'wild_places_b1':     ('wild-places', 7, 1, [(4096, 'ball')]),
    'wild_places_ragged': ('wild-places', 7, 11, [(4096, 'ball'), (60, 'ball'), (700, 'forest'),
                                                  (9, 'ball'), (4096, 'forest')]),
    'cs_wild_places_b2':  ('cs-wild-places', 7, 3, [(6000, 'forest'), (4096, 'ball')]),
    'oxford_b2':          ('oxford', 9, 5, [(4096, 'ball')] * 2),
    'cs_campus3d_b2':     ('cs-campus3d', 7, 7, [(4096, 'forest'), (5000, 'ball')]),
}


# The BASELINE.json workloads themselves, at their full sizes: clouds from `bench.py::bench_clouds` and the bench's weight
# profile.
#   case -> (cfg, octree depth, weight profile, make_clouds config id, batch, n_points, n_points_max or None)
WORKLOAD_CASES = {
    'wild_places_b32':       ('wild-places', 7, 'init', 2, 32, 4096, None),        # config 2 = the bench's timed batch
    'cs_wild_places_b8_var': ('cs-wild-places', 7, 'init', 3, 8, 4096, 32768),     # config 3's generator, 8 clouds
    'cs_wild_places_b64_var': ('cs-wild-places', 7, 'init', 3, 64, 4096, 32768),   # config 3 itself: 64 clouds, 1.1 M points
}
# cases whose fixture does not carry its points (13 MB for config 3): cartesian clouds come out of integer hashes and exactly
# rounded arithmetic (hotformerloc_amd/synthetic.py), the fixture keeps their SHA-256 and the per-depth node counts so that
# a test can tell "different input" from "different result"
DESCRIPTORS_ONLY = {'cs_wild_places_b64_var'}


def workload_clouds(coordinates, cid, batch, n_points, n_points_max):
    """bench.py::bench_clouds for rank 0"""
    if n_points_max:
        clouds = []
        for i in range(batch):
            clouds += syn.make_clouds(cid, 1, n_points, coordinates, kind='forest' if i % 2 == 0 else 'ball',
                                      n_points_max=n_points_max, first_index=i)
        return clouds
    return syn.make_clouds(cid, batch, n_points, coordinates)


def build_case(case):
    workload = case in WORKLOAD_CASES
    if workload:
        cfg, depth, profile, cid, batch, n_points, n_points_max = WORKLOAD_CASES[case]
    else:
        cfg, depth, cid, spec = CASES[case]
        profile = 'stress'
    cfg_path = os.path.join(ref_import.REFERENCE_ROOT, 'models', 'hotformerloc_%s_cfg.txt' % cfg)
    model, params = ref_import.reference_model(cfg_path)
    syn.fill_synthetic_weights(model, profile)
    clouds = []
    if workload:
        clouds = workload_clouds(params.coordinates, cid, batch, n_points, n_points_max)
    else:
      for i, (n, kind) in enumerate(spec):
        seed = 1000 * cid + i
        pc = syn.unit_ball_cloud(seed, n) if kind == 'ball' else syn.forest_cloud(seed, n)
        if params.coordinates == 'cylindrical':
            pc = syn.cylindrical(pc)
        clouds.append(pc)
    octs = []
    for pc in clouds:
        o = Octree(depth, 2)
        o.build_octree(Points(torch.from_numpy(pc)))
        octs.append(o)
    octree = merge_octrees(octs)
    octree.construct_all_neigh()

    cap = {}
    base = model.backbone.backbone
    hooks = [
        base.patch_embed.register_forward_hook(lambda m, i, o: cap.__setitem__('patch_embed', o)),
        base.downsample[0].register_forward_hook(lambda m, i, o: cap.__setitem__('octf_out', o)),
        base.hotf_stage.register_forward_hook(lambda m, i, o: cap.__setitem__('hotf', o)),
    ]
    with torch.inference_mode():
        y = model({'octree': octree})['global']
    for h in hooks:
        h.remove()
    assert torch.isfinite(y).all()

    out = dict(cfg=np.array(cfg), octree_depth=np.array(depth),
               n_points=np.array([c.shape[0] for c in clouds], dtype=np.int64),
               descriptors=y.numpy().astype(np.float32),
               nnum_nempty=octree.nnum_nempty.numpy())
    # (the points themselves, also for the generated workloads: the cylindrical transform's float64 chain is not
    # bit-reproducible across host CPUs, and one point that changes its depth-7 cell changes the octree)
    pts = np.ascontiguousarray(np.concatenate(clouds, 0).astype(np.float32))
    if case in DESCRIPTORS_ONLY:
        import hashlib
        out['points_sha256'] = np.array(hashlib.sha256(pts.tobytes()).hexdigest())
    else:
        out['points'] = pts
    if workload:      # the generator's arguments (bench.py::bench_clouds) and the weight profile
        out['workload'] = np.array([cid, batch, n_points, n_points_max or 0], dtype=np.int64)
        out['profile'] = np.array(profile)

    def put(name, t):
        t = t.detach().double()
        if case not in DESCRIPTORS_ONLY:
            out[name + '_head'] = t[:32].float().numpy()
        out[name + '_sum'] = np.array([t.sum().item(), (t * t).sum().item()])
    put('patch_embed', cap['patch_embed'])
    put('octf_out', cap['octf_out'])
    feats, rts = cap['hotf']
    for d in feats:
        put('feat_final_%d' % d, feats[d])
        put('rt_final_%d' % d, rts[d])
    return out


def dump_state_dicts(dst):
    """tests/golden/state_dict_<cfg>.json: [[name, shape], ...] of the reference model's state_dict, in order
    (tests/test_host_logic.py::test_state_dict_matches_reference_layout)."""
    import json
    for cfg in ('wild-places', 'cs-wild-places', 'oxford', 'cs-campus3d'):
        cfg_path = os.path.join(ref_import.REFERENCE_ROOT, 'models', 'hotformerloc_%s_cfg.txt' % cfg)
        model, _ = ref_import.reference_model(cfg_path)
        spec = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        with open(os.path.join(dst, 'state_dict_%s.json' % cfg.replace('-', '_')), 'w') as f:
            json.dump(spec, f)
        print('state_dict', cfg, len(spec), 'tensors', sum(p.numel() for p in model.parameters()), 'parameters')


def main():
    dst = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(dst, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    if sys.argv[1:] == ['state_dicts']:
        dump_state_dicts(dst)
        return
    for case in (sys.argv[1:] or list(CASES) + list(WORKLOAD_CASES)):
        out = build_case(case)
        path = os.path.join(dst, 'model_%s.npz' % case)
        np.savez_compressed(path, **out)
        d = out['descriptors']
        print(case, out['n_points'].tolist(), 'nne', out['nnum_nempty'].tolist(),
              'desc[0,:3]', d[0, :3], 'pairwise', np.round((d @ d.T)[:4, :4], 3).tolist(),
              '%.0f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
