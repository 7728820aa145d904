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
for depth in (3, 4):
    nt, W = plan.n_tokens[depth], plan.n_windows[depth]
    for seed in (3, 4):
        g = torch.Generator().manual_seed(seed)
        q = torch.randn(nt + W, 3 * C, generator=g); q[:, 2 * C:] = 1.0
        got = ops.window_attention(_pack_qkv_f16(q, H, 0.25 * 1.4426950408889634).cuda(), plan.meta[depth], None, nt, W, K, 1, G, H, 2,
                                   rt_row0=nt, depth=depth, qkv_f16=True).cpu()
        r = got[nt:].view(W, H, 16)
        bad = ((r - 1).abs() > 1e-3).any(2)
        pairs = [(int(w), int(h), round(float(r[w, h, 0]), 3)) for w, h in bad.nonzero().tolist()]
        print('depth', depth, 'seed', seed, 'W', W, 'bad (window, head, value):', pairs[:24])
        # relay q.k score vs token scores for the bad pairs: is the relay key the arg-max?
        qq = q[nt:, :C].view(W, H, 16); kk = q[nt:, C:2 * C].view(W, H, 16)
        for w, h, _ in pairs[:6]:
            srr = (qq[w, h] * kk[w, h]).sum().item() * 0.25
            kt = q[w * K:(w + 1) * K, C:2 * C].view(-1, H, 16)[:, h]
            st = (kt @ qq[w, h]) * 0.25
            print('    (w %d, h %d): relay-relay score %.2f, token scores max %.2f min %.2f' % (w, h, srr, st.max().item(), st.min().item()))
