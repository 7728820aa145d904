"""The window-attention backward at ONE shape, a few launches (for counter surveys): CS-Wild-Places cfg, B = 64, 8192 points per
cloud, the depth-4 level (K = 64 + relay token, 16 heads).  tools/attn_bwd_one.py [depth offset from the finest level: 0]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn, _native, autograd as ag
from hotformerloc_amd.plan import WindowPlan
params, depth = load_config('cs-wild-places')
B = 64
octree = build_batch_octree(syn.make_clouds(2, B, 8192, params.coordinates), depth, 2, 'cuda')
md = depth - 2
plan = WindowPlan(octree, params.patch_size, params.dilation, md, md - 3, 3, 1, params.ADaPE_mode)
K = params.patch_size
d = md - 1 - (int(sys.argv[1]) if len(sys.argv) > 1 else 0)
H, G, C = 16, 1, 256
lib = _native.load()
g = torch.Generator(device='cuda').manual_seed(0)
nt, W = plan.n_tokens[d], plan.n_windows[d]
rows = nt + W
qkv = torch.randn(rows, 3 * C, device='cuda', generator=g)
dout = torch.randn(rows, C, device='cuda', generator=g)
bnd = int(0.8 * K)
table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.1
desc = ag._desc(nt, W, K, 1, G, H, B, nt, d)
dqkv = torch.zeros_like(qkv); dtab = torch.zeros_like(table)
for _ in range(10):
    ag._window_attention_bwd(dqkv, dtab, qkv, dout, plan.meta[d], table, desc)
torch.cuda.synchronize()
print('depth', d, 'rows', rows, 'windows', W)
