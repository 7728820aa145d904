import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from hotformerloc_amd import ops, synthetic as syn, build_batch_octree, load_config, _native
from hotformerloc_amd.plan import WindowPlan
from test_gpu_kernels import _pack_qkv_f16
lib = _native.load()
params, _ = load_config('cs-wild-places')
clouds = [syn.unit_ball_cloud(500 + i, n) for i, n in enumerate([5000, 3000])]
dev = build_batch_octree(clouds, 7, 2, 'cuda')
plan = WindowPlan(dev, params.patch_size, params.dilation, 5, 2, 3, 1, params.ADaPE_mode)
K = params.patch_size
g = torch.Generator().manual_seed(3)
depth = 3; H, G, C = 16, 1, 256
nt, W = plan.n_tokens[depth], plan.n_windows[depth]
qkv = torch.randn(nt + W, 3 * C, generator=g)
pk = _pack_qkv_f16(qkv, H, 0.25 * 1.4426950408889634).cuda()
want = ops.window_attention(qkv.cuda(), plan.meta[depth], None, nt, W, K, 1, G, H, 2, rt_row0=nt, depth=depth).cpu()
for hpw, wgs in ((4, 1), (4, 0), (2, 1)):
    lib.hfl_set_variant(b'window_heads_per_wg', hpw)
    lib.hfl_set_variant(b'window_v4_wgs_per_cu', wgs)
    outs = []
    for rep in range(3):
        got = ops.window_attention(pk, plan.meta[depth], None, nt, W, K, 1, G, H, 2, rt_row0=nt, depth=depth, qkv_f16=True).cpu()
        outs.append(got)
    e = (outs[0][nt:] - want[nt:]).abs().amax(1)
    print('hpw', hpw, 'wgs', wgs, 'relay err per window', ['%.1e' % v for v in e.tolist()], 'deterministic', torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]))
    # which heads / channels are wrong in the worst relay row
    wbad = int(e.argmax())
    d = (outs[0][nt + wbad] - want[nt + wbad]).abs().view(H, 16)
    print('   worst window', wbad, 'bad heads', (d.amax(1) > 1e-3).nonzero().flatten().tolist(), 'per-channel of first bad head', ['%.1e' % v for v in d[(d.amax(1) > 1e-3).nonzero().flatten()[0]].tolist()] if (d.amax(1) > 1e-3).any() else None)
lib.hfl_set_variant(b'window_heads_per_wg', 4)
lib.hfl_set_variant(b'window_v4_wgs_per_cu', 1)
