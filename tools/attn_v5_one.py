"""Only the fp16 (hi, lo) window-attention kernel at the bench's depth-4 shape, a few launches (rocprofv3 --pmc surveys)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn  # noqa: E402
from hotformerloc_amd.plan import WindowPlan  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
params, depth = load_config('wild-places')
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
plan = WindowPlan(octree, 48, 4, 5, 2, 3, 1, None)
g = torch.Generator(device='cuda').manual_seed(0)
H, G, C = 16, 1, 256
nt, W = plan.n_tokens[d], plan.n_windows[d]
rows = nt + W
x = torch.randn(rows, C, device='cuda', generator=g)
w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.06
b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
qkv = ops.linear_x3_qkv(ops.split2(x), ops.split2_weight(w), b, 16 ** -0.5 * 1.4426950408889634)
table = torch.randn(3 * 77, H, device='cuda', generator=g) * 0.1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.window_attention(qkv, plan.meta[d], table, nt, W, 48, 1, G, H, 32, rt_row0=nt, depth=d, out_split=2, qkv_f16=True)
torch.cuda.synchronize()
e0.record()
for _ in range(n):
    ops.window_attention(qkv, plan.meta[d], table, nt, W, 48, 1, G, H, 32, rt_row0=nt, depth=d, out_split=2, qkv_f16=True)
e1.record()
torch.cuda.synchronize()
print('depth %d rows %d: %.1f us per launch, %.0f GB/s algorithmic' % (d, rows, e0.elapsed_time(e1) / n * 1e3,
                                                                        rows * C * 16 / (e0.elapsed_time(e1) / n * 1e-3) / 1e9))
