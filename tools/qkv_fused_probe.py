"""hfl_ln_qkv_fused against hfl_layer_norm_split2 + hfl_linear_x3_qkv: decoded values and timing at the bench's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops  # noqa: E402


def decode(buf, c):
    """(rows, 3C) operand buffer -> (rows, 3C) float32: per head 64 B = [16 x hi | 16 x lo] fp16"""
    rows = buf.shape[0]
    h = buf.view(torch.float16).reshape(rows, 3, c // 16, 2, 16).float()
    return (h[:, :, :, 0] + h[:, :, :, 1]).reshape(rows, 3 * c)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator(device='cuda').manual_seed(0)
for rows, c in ((66775, 256), (13984, 256), (300, 256), (118096, 128), (65536, 256), (131072, 128), (17, 128)):
    x = torch.randn(rows, c, device='cuda', generator=g) * 1.5 + 0.3
    gamma = torch.rand(c, device='cuda', generator=g) + 0.5
    beta = torch.randn(c, device='cuda', generator=g) * 0.1
    w = torch.randn(3 * c, c, device='cuda', generator=g) * 0.06
    b = torch.randn(3 * c, device='cuda', generator=g) * 0.1
    qs = 16 ** -0.5 * 1.4426950408889634
    w2 = ops.split2_weight(w)
    pack = ops.qkv_fused_pack(w)

    def unfused():
        return ops.linear_x3_qkv(ops.layer_norm_split2(x, gamma, beta, 1e-5), w2, b, qs)

    def fused():
        return ops.ln_qkv_fused(x, gamma, beta, 1e-5, pack, b, qs)
    a, f = decode(unfused(), c), decode(fused(), c)
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x.double(), (c,), gamma.double(), beta.double(), 1e-5),
                                     w.double(), b.double())
    ref[:, :c] *= qs
    err_f = ((f.double() - ref).norm() / ref.norm()).item()
    err_a = ((a.double() - ref).norm() / ref.norm()).item()
    tu, tf = timeit(unfused), timeit(fused)
    print('rows %6d C %3d: fused %7.1f us  LN + qkv %7.1f us  x%.2f | rel err vs f64: fused %.2e  unfused %.2e  max |fused - unfused| %.2e'
          % (rows, c, tf, tu, tu / tf, err_f, err_a, (f - a).abs().max().item()), flush=True)
