"""hfl_gemm_bf16 (hipBLASLt called directly, bias + residual epilogue) vs the torch.mm route."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import ops

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

g = torch.Generator(device='cuda').manual_seed(0)
for M, K, N in ((68167, 256, 768), (68167, 256, 256), (68167, 256, 1024), (68167, 1024, 256), (14276, 1024, 256),
                (2092, 256, 1024), (118096, 128, 384), (118096, 512, 128), (77, 256, 256)):
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) * 0.05
    b = torch.randn(N, device='cuda', generator=g)
    r = torch.randn(M, N, device='cuda', generator=g)
    a3 = ops.split3(x); w3 = ops.split_weight(w)
    ref = (x.double() @ w.double().t())
    y0 = ops.split_mm(a3, w3)
    y1 = ops.gemm_bf16(a3, w3)
    y2 = ops.gemm_bf16(a3, w3, bias=b, residual=r)
    y3 = ops.gemm_bf16(a3, w3, bias=b)
    sc = ref.abs().max()
    e0 = ((y0 - ref).abs().max() / sc).item(); e1 = ((y1 - ref).abs().max() / sc).item()
    e2 = ((y2 - (ref + b + r)).abs().max() / sc).item(); e3 = ((y3 - (ref + b)).abs().max() / sc).item()
    t0 = t(lambda: ops.split_mm(a3, w3)); t1 = t(lambda: ops.gemm_bf16(a3, w3))
    t2 = t(lambda: ops.gemm_bf16(a3, w3, bias=b, residual=r))
    t3 = t(lambda: ops.add_bias(r, ops.split_mm(a3, w3), b))
    print('M=%6d K=%4d N=%4d err mm %.1e lt %.1e lt+bias+res %.1e lt+bias %.1e | torch.mm %.1f us, lt %.1f us, lt+bias+res %.1f us, '
          'torch.mm + add_bias kernel %.1f us' % (M, K, N, e0, e1, e2, e3, t0, t1, t2, t3))
