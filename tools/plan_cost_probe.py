"""Host + device cost of what `model(batch)` derives from a resident octree per forward (bench.py's boundary step):
live-tap lists (one device->host read), window plan pieces.  python tools/plan_cost_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, model_factory  # noqa: E402
from hotformerloc_amd import synthetic as syn  # noqa: E402
from hotformerloc_amd.plan import WindowPlan, window_layout  # noqa: E402

params, depth = load_config('wild-places')
clouds = syn.make_clouds(2, 32, 4096, params.coordinates)
octree = build_batch_octree(clouds, depth, 2, 'cuda')
model = model_factory(params)
syn.fill_synthetic_weights(model, 'init')
model = model.cuda().eval()
base = model.backbone.backbone


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def taps():
    octree.__dict__.pop('_sparse_taps', None)
    octree.__dict__.pop('_tap_tiles', None)
    octree.construct_all_neigh()


def tiles():
    octree.__dict__.pop('_tap_tiles', None)
    for d, k, s, w in ((6, '333', 1, 128), (5, '333', 1, 128), (7, '222', 2, 128), (6, '222', 2, 128), (5, '222', 2, 256),
                       (4, '222', 2, 256), (3, '222', 2, 256)):
        try:
            octree.tap_tiles(d, k, s, w)
        except Exception as e:        # noqa
            pass


def plan():
    octree.__dict__.pop('_window_plans', None)
    WindowPlan.for_octree(octree, base.patch_size, base.dilation, max_depth=depth - base.stem_down,
                          start_depth=depth - base.stem_down - base.num_stages + 1, num_pyramid_levels=base.num_pyramid_levels,
                          num_octf_levels=base.num_octf_levels, adape_mode=base.ADaPE_mode)


def layout_only():
    window_layout(octree.batch_nnum_nempty.numpy(), base.patch_size, base.dilation, depth - base.stem_down,
                  depth - base.stem_down - base.num_stages + 1, [depth - base.stem_down - base.num_octf_levels - j
                                                                 for j in range(base.num_pyramid_levels)])


with torch.inference_mode():
    def fwd_resident():
        model({'octree': octree})

    def fwd_boundary():
        octree.drop_forward_caches()
        model({'octree': octree})
    print('tap lists (kernels + 1 host read)   %.3f ms' % t(taps))
    print('tap tile tables (host numpy + H2D)  %.3f ms' % t(tiles))
    print('window plan (all)                   %.3f ms' % t(plan))
    print('  window_layout (host numpy)        %.3f ms' % t(layout_only))
    print('forward, resident plan              %.3f ms' % t(fwd_resident))
    print('forward, boundary (caches dropped)  %.3f ms' % t(fwd_boundary))
