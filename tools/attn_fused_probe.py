"""hfl_attn_fused_fwd against hfl_ln_qkv_fused + the fp16 window kernel at the bench's depth-5 shape: equality and timing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn  # noqa: E402
from hotformerloc_amd.plan import WindowPlan  # noqa: E402


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


params, depth = load_config('wild-places')
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
plan = WindowPlan(octree, 48, 4, 5, 2, 3, 1, None)
d, H, C, K = 5, 8, 128, 48
nt = plan.n_tokens[d]
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(nt, C, device='cuda', generator=g)
gamma = torch.rand(C, device='cuda', generator=g) + 0.5
beta = torch.randn(C, device='cuda', generator=g) * 0.1
w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.06
b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
qs = 16 ** -0.5 * 1.4426950408889634
pack = ops.qkv_fused_pack(w)
for dil in (1, 4):
    W = -(-nt // (K * dil)) * dil
    bnd = int(0.8 * K * dil ** 0.5)
    table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.1

    def two():
        qkv = ops.ln_qkv_fused(x, gamma, beta, 1e-5, pack, b, qs)
        return ops.window_attention(qkv, plan.meta[d], table, nt, W, K, dil, 0, H, plan.B, rt_row0=nt, depth=d, out_split=2,
                                    qkv_f16=True)

    def one():
        return ops.attn_fused(x, gamma, beta, 1e-5, pack, b, qs, plan.meta[d], table, nt, W, K, dil, H, plan.B, d)
    a, f = two(), one()
    same = torch.equal(a[:nt].view(torch.int16), f[:nt].view(torch.int16))
    nbad = (a[:nt].view(torch.int16) != f[:nt].view(torch.int16)).any(dim=1).sum().item()
    t2, t1 = timeit(two), timeit(one)
    flop = (6.0 * nt * C * C * 3 + 4.0 * K * K * C * (nt / K) * 3.5)
    print('depth 5 rows %d dilation %d: fused %.1f us  two launches %.1f us  x%.2f | bitwise equal %s (%d rows differ) | '
          '%.0f TF/s issued (bf16 + fp16 MFMA), %.2f TB/s of x + out' % (nt, dil, t1, t2, t2 / t1, same, nbad, flop / t1 / 1e6,
                                                                       nt * C * 8 / t1 / 1e6), flush=True)
