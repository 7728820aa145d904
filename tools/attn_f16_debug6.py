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
H, G, C = 16, 1, 256
depth = 3
nt, W = plan.n_tokens[depth], plan.n_windows[depth]
g = torch.Generator().manual_seed(3)
tok = torch.randn(nt, 3 * C, generator=g); rel = torch.randn(W, 3 * C, generator=g)
q = torch.cat([tok, rel]).clone(); q[:, 2 * C:] = 1.0
lib.hfl_set_variant(b'window_debug', 64)
got = ops.window_attention(_pack_qkv_f16(q, H, 0.25 * 1.4426950408889634).cuda(), plan.meta[depth], None, nt, W, K, 1, G, H, 2,
                           rt_row0=nt, depth=depth, qkv_f16=True).cpu()
lib.hfl_set_variant(b'window_debug', 0)
torch.set_printoptions(linewidth=250, precision=4, sci_mode=False)
d = got[:64, :28]
print('lane: mx sum inv srt | o0..3 | s[0] s[1] s[2] s[3] s[4]  (P after exp for kt<4, s[4]={ert,0,0,0})')
for lane in (0, 16, 32, 48, 1, 17):
    print(lane, d[lane].tolist())

print('final relay row (w=2) head 0:', got[nt + 2, :16].tolist())
# expected scores of the relay query (w=2, h=0) against its 64 token keys and itself, exp2 domain
qq = q[nt + 2, :16] * 0.25 * 1.4426950408889634
kt = q[2 * K:3 * K, C:C + 16]
print('expected token scores (exp2 domain): max %.3f min %.3f; relay-relay %.3f' % ((kt @ qq).max().item(), (kt @ qq).min().item(), (q[nt + 2, C:C + 16] @ qq).item()))
