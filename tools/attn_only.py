"""Run only the window-attention kernel at bench shapes (for rocprofv3 --pmc runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn
from hotformerloc_amd.plan import WindowPlan
params, depth = load_config('wild-places')
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
plan = WindowPlan(octree, 48, 4, 5, 2, 3, 1, None)
g = torch.Generator(device='cuda').manual_seed(0)
split = 'split' in sys.argv[1:]
from hotformerloc_amd import _native
for a in sys.argv[1:]:
    if a.startswith('v'):
        _native.load().hfl_set_variant(b'window_attention', int(a[1:]))
for d, H, G in ((4, 16, 1),):
    C = H * 16
    nt, W = plan.n_tokens[d], plan.n_windows[d]
    qkv = torch.randn(nt + (W if G else 0), 3 * C, device='cuda', generator=g)
    table = torch.randn(3 * 77, H, device='cuda', generator=g) * 0.1
    bias = torch.randn(3 * C, device='cuda', generator=g)
    for _ in range(5):
        ops.window_attention(qkv, plan.meta[d], table, nt, W, 48, 1, G, H, 32, rt_row0=nt, depth=d,
                             qkv_bias=bias if split else None, out_split=split)
torch.cuda.synchronize()
