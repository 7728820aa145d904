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
g = torch.Generator().manual_seed(3)
depth = 3; H, G, C = 16, 1, 256
nt, W = plan.n_tokens[depth], plan.n_windows[depth]
base = torch.randn(nt + W, 3 * C, generator=g)
def run(qkv, tag):
    want = ops.window_attention(qkv.cuda(), plan.meta[depth], None, nt, W, K, 1, G, H, 2, rt_row0=nt, depth=depth).cpu()
    got = ops.window_attention(_pack_qkv_f16(qkv, H, 0.25 * 1.4426950408889634).cuda(), plan.meta[depth], None, nt, W, K, 1, G, H, 2,
                               rt_row0=nt, depth=depth, qkv_f16=True).cpu()
    e = (got[nt:] - want[nt:]).abs().amax(1)
    print(tag, 'relay err per window', ['%.1e' % v for v in e.tolist()])
    return got, want
q = base.clone(); run(q, 'random         ')
q = base.clone(); q[:, 2 * C:] = 1.0; got, want = run(q, 'v = 1          ')
print('   relay rows v=1 got', got[nt:nt + 8, 128].tolist())
q = base.clone(); q[nt:, :C] = 0.0; run(q, 'relay q = 0    ')          # uniform attention for the relay query
q = base.clone(); q[nt:, C:2 * C] = 0.0; run(q, 'relay k = 0    ')
q = base.clone(); q[nt:, 2 * C:] = 0.0; run(q, 'relay v = 0    ')
q = base.clone(); q[:nt, 2 * C:] = 0.0; run(q, 'token v = 0    ')
print('---- sentinel test: which relay (window, head) entries are never written')
import hotformerloc_amd.ops as O
real_empty = torch.empty
def fake_empty(shape, **kw):
    return torch.full(shape, 7.0, **kw) if kw.get('dtype') == torch.float32 else real_empty(shape, **kw)
O.torch.empty = fake_empty
q = base.clone(); q[:, 2 * C:] = 1.0
got = ops.window_attention(_pack_qkv_f16(q, H, 0.25 * 1.4426950408889634).cuda(), plan.meta[depth], None, nt, W, K, 1, G, H, 2,
                           rt_row0=nt, depth=depth, qkv_f16=True).cpu()
O.torch.empty = real_empty
r = got[nt:].view(W, H, 16)
for w in range(W):
    print('window', w, 'heads untouched (== 7.0):', (r[w] == 7.0).all(1).nonzero().flatten().tolist(),
          'heads == 1:', ((r[w] - 1).abs() < 1e-3).all(1).sum().item(), 'other values sample', r[w][((r[w] != 7.0) & ((r[w] - 1).abs() > 1e-3)).any(1)][:1, :4].tolist())
print('token rows untouched:', (got[:nt] == 7.0).any(1).sum().item())
