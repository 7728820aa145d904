import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from hotformerloc_amd import ops, synthetic as syn, build_batch_octree, load_config
from hotformerloc_amd.plan import WindowPlan
from test_gpu_kernels import _pack_qkv_f16
params, _ = load_config('cs-wild-places')
clouds = [syn.unit_ball_cloud(500 + i, n) for i, n in enumerate([5000, 3000])]
dev = build_batch_octree(clouds, 7, 2, 'cuda')
plan = WindowPlan(dev, params.patch_size, params.dilation, 5, 2, 3, 1, params.ADaPE_mode)
K = params.patch_size
g = torch.Generator().manual_seed(3)
for depth in (4, 3, 2):
    H, G = 16, 1; C = 256
    nt, W = plan.n_tokens[depth], plan.n_windows[depth]
    qkv = torch.randn(nt + W, 3 * C, generator=g)
    table = torch.randn(3 * (2 * int(0.8 * K) + 1), H, generator=g) * 0.5
    want = ops.window_attention(qkv.cuda(), plan.meta[depth], table.cuda(), nt, W, K, 1, G, H, 2, rt_row0=nt, depth=depth).cpu()
    got = ops.window_attention(_pack_qkv_f16(qkv, H, 0.25 * 1.4426950408889634).cuda(), plan.meta[depth], table.cuda(), nt, W, K, 1, G, H, 2,
                               rt_row0=nt, depth=depth, qkv_f16=True).cpu()
    bad = torch.isnan(got)
    print('depth', depth, 'nt', nt, 'W', W, 'nan rows', bad.any(1).nonzero().flatten().tolist()[:20], 'nan cols of first bad row',
          bad[bad.any(1).nonzero().flatten()[0]].nonzero().flatten().tolist()[:40] if bad.any() else None)
    ok = ~bad.any(1)
    print('   max err on finite rows', (got[ok] - want[ok]).abs().max().item(), 'relay rows', list(range(nt, nt + W))[:5], '...')
print('---- per-window error map, depth 3')
depth = 3; H, G, C = 16, 1, 256
nt, W = plan.n_tokens[depth], plan.n_windows[depth]
bid = plan.meta[depth][:, 1].cpu()
for tbl_on in (True, False):
    qkv = torch.randn(nt + W, 3 * C, generator=g)
    table = torch.randn(3 * (2 * int(0.8 * K) + 1), H, generator=g) * 0.5
    t = table.cuda() if tbl_on else None
    want = ops.window_attention(qkv.cuda(), plan.meta[depth], t, nt, W, K, 1, G, H, 2, rt_row0=nt, depth=depth).cpu()
    got = ops.window_attention(_pack_qkv_f16(qkv, H, 0.25 * 1.4426950408889634).cuda(), plan.meta[depth], t, nt, W, K, 1, G, H, 2,
                               rt_row0=nt, depth=depth, qkv_f16=True).cpu()
    err = (got - want).abs().nan_to_num(99.0)
    for w in range(W):
        rows = slice(w * K, min((w + 1) * K, nt))
        e_tok = err[rows].max().item() if w * K < nt else 0.0
        per_head = err[rows].view(-1, H, 16).amax((0, 2)) if w * K < nt else torch.zeros(H)
        b = bid[rows]
        print('rpe', tbl_on, 'window', w, 'tokens', b.numel(), 'homog', bool((b == b[0]).all()) if b.numel() else None,
              'tok err %.2e' % e_tok, 'relay err %.2e' % err[nt + w].max().item(),
              'bad heads', (per_head > 1e-3).nonzero().flatten().tolist(),
              'bad token offsets', (err[rows].amax(1) > 1e-3).nonzero().flatten().tolist()[:12])
