"""Window attention at the bench shapes: fp32-qkv kernel (v4) vs fp16 (hi, lo) operand kernel (v5), back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn
from hotformerloc_amd.plan import WindowPlan
cfg = sys.argv[1] if len(sys.argv) > 1 else 'wild-places'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
params, depth = load_config(cfg)
octree = build_batch_octree(syn.make_clouds(2, B, 4096, params.coordinates), depth, 2, 'cuda')
md = depth - 2
plan = WindowPlan(octree, params.patch_size, params.dilation, md, md - 3, 3, 1, params.ADaPE_mode)
K = params.patch_size
def timeit(fn, rounds=10, inner=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]
g = torch.Generator(device='cuda').manual_seed(0)
for d, H, G, dil in ((md, 8, 0, 1), (md, 8, 0, params.dilation), (md - 1, 16, 1, 1), (md - 2, 16, 1, 1), (md - 3, 16, 1, 1)):
    C = H * 16
    nt, W = plan.n_tokens[d], plan.n_windows[d]
    rows = nt + (W if G else 0)
    qkv = torch.randn(rows, 3 * C, device='cuda', generator=g)
    bnd = int(0.8 * K * dil ** 0.5)
    table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.1
    t4 = timeit(lambda: ops.window_attention(qkv, plan.meta[d], table, nt, W, K, dil, G, H, B, rt_row0=nt, depth=d, out_split=2))
    ok = ops.window_attention_f16_ok(rows, K, dil, G, H, d)
    t5 = timeit(lambda: ops.window_attention(qkv, plan.meta[d], table, nt, W, K, dil, G, H, B, rt_row0=nt, depth=d, out_split=2, qkv_f16=True)) if ok else float('nan')
    print('%s d=%d H=%d G=%d D=%d rows %d: fp32 qkv %.1f us (%.0f GB/s)   fp16 operands %.1f us (%.0f GB/s)  eligible %s'
          % (cfg, d, H, G, dil, rows, t4, rows * C * 16 / t4 / 1e3, t5, rows * C * 16 / t5 / 1e3 if ok else 0, ok))
