import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from hotformerloc_amd import ops, synthetic as syn, build_batch_octree, load_config, _native
from hotformerloc_amd.plan import WindowPlan
from test_gpu_kernels import _pack_qkv_f16
params, _ = load_config('cs-wild-places')
clouds = [syn.unit_ball_cloud(500 + i, n) for i, n in enumerate([5000, 3000])]
dev = build_batch_octree(clouds, 7, 2, 'cuda')
plan = WindowPlan(dev, params.patch_size, params.dilation, 5, 2, 3, 1, params.ADaPE_mode)
K = params.patch_size
H, G, C = 16, 1, 256
depth = 3
nt, W = plan.n_tokens[depth], plan.n_windows[depth]
def bad_pairs(q):
    got = ops.window_attention(_pack_qkv_f16(q, H, 0.25 * 1.4426950408889634).cuda(), plan.meta[depth], None, nt, W, K, 1, G, H, 2,
                               rt_row0=nt, depth=depth, qkv_f16=True).cpu()
    r = got[nt:].view(W, H, 16)
    bad = ((r - 1).abs() > 1e-3).any(2)
    return [(int(w), int(h), round(float(r[w, h, 0]), 3)) for w, h in bad.nonzero().tolist()]
g = torch.Generator().manual_seed(3)
tok = torch.randn(nt, 3 * C, generator=g); rel_a = torch.randn(W, 3 * C, generator=g); rel_b = torch.randn(W, 3 * C, generator=g)
tok2 = torch.randn(nt, 3 * C, generator=g)
for name, t, r in (('tok1+relA', tok, rel_a), ('tok1+relB', tok, rel_b), ('tok2+relA', tok2, rel_a)):
    q = torch.cat([t, r]).clone(); q[:, 2 * C:] = 1.0
    print(name, bad_pairs(q))
q = torch.cat([tok, rel_a]).clone(); q[:, 2 * C:] = 1.0; q[:nt, :C] = 0.0
print('token q = 0      ', bad_pairs(q))
q = torch.cat([tok, rel_a]).clone(); q[:, 2 * C:] = 1.0; q[:nt, C:2 * C] = 0.0
print('token k = 0      ', bad_pairs(q))
q = torch.cat([tok, rel_a]).clone(); q[:, 2 * C:] = 1.0; q[:nt, :2 * C] = 0.0; q[nt:, :2 * C] = 0.0
print('all q = k = 0    ', bad_pairs(q))
