"""Only the fp16 (hi, lo) window-attention kernel at the bench's four launch shapes (depth 5: C = 128, no relay row; depths
4, 3, 2: C = 256 with the relay row), back to back -- for rocprofv3 --pmc surveys and A/B runs of kernel changes:
`HFL_LIB=<other .so> python tools/attn_v5_one.py` loads another build of the library; the output sums let two builds be
compared for equality.  argv: [depths, e.g. 5,4,3,2] [launches per shape]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import _native  # noqa: E402

if os.environ.get('HFL_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['HFL_LIB'])
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn  # noqa: E402
from hotformerloc_amd.plan import WindowPlan  # noqa: E402

for kv in os.environ.get('HFL_KNOBS', '').split(','):          # e.g. HFL_KNOBS=window_rpe_form1_max_depth=5,window_v4_wgs_per_cu=2
    if '=' in kv:
        _native.load().hfl_set_variant(kv.split('=')[0].encode(), int(kv.split('=')[1]))
depths = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [5, 4, 3, 2]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
params, depth = load_config('wild-places')
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
plan = WindowPlan(octree, 48, 4, 5, 2, 3, 1, None)
g = torch.Generator(device='cuda').manual_seed(0)
total_us = 0.0
total_b = 0.0
for d in depths:
    H, G, C = (8, 0, 128) if d == 5 else (16, 1, 256)
    nt, W = plan.n_tokens[d], plan.n_windows[d]
    rows = nt + W * G
    x = torch.randn(rows, C, device='cuda', generator=g)
    w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.06
    b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
    qkv = ops.linear_x3_qkv(ops.split2(x), ops.split2_weight(w), b, 16 ** -0.5 * 1.4426950408889634)
    table = None if os.environ.get('HFL_NO_RPE') else torch.randn(3 * 77, H, device='cuda', generator=g) * 0.1

    def run():
        return ops.window_attention(qkv, plan.meta[d], table, nt, W, 48, 1, G, H, 32, rt_row0=nt, depth=d, out_split=2,
                                    qkv_f16=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        y = run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    yv = y.view(torch.int16).to(torch.int64)
    total_us += us * (4 if d == 5 else 10)
    total_b += rows * C * 16 * (4 if d == 5 else 10)
    print('depth %d rows %6d: %6.1f us per launch, %5.0f GB/s algorithmic (%.3f of 8 TB/s)   out checksum %d' % (
        d, rows, us, rows * C * 16 / (us * 1e-6) / 1e9, rows * C * 16 / (us * 1e-6) / 8e12, int((yv * yv % 1000003).sum())),
        flush=True)
if len(depths) == 4:
    print('step-weighted (4 x depth 5 + 10 x the others): %.1f us per launch over 34, frac %.3f' % (
        total_us / 34, total_b / (total_us * 1e-6) / 8e12))
