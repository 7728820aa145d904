"""Run only the fused CPE kernel at bench shapes (for rocprofv3 --pmc runs): depth 4, C = 256."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn
from hotformerloc_amd.plan import WindowPlan
params, depth = load_config('wild-places')
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
plan = WindowPlan(octree, 48, 4, 5, 2, 3, 1, None)
g = torch.Generator(device='cuda').manual_seed(0)
d, C = 4, 256
n = plan.n_tokens[d]
x = torch.randn(n, C, device='cuda', generator=g)
w = torch.randn(27, 1, C, device='cuda', generator=g)
gm = torch.ones(C, device='cuda'); bt = torch.zeros(C, device='cuda')
neigh = plan.neigh(d)
for _ in range(5):
    ops.cpe_forward(x, w, gm, bt, neigh, True)
torch.cuda.synchronize()
print('rows', n, 'live taps per row', float((neigh >= 0).float().sum(1).mean()))
