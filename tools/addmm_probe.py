"""Does torch.addmm(residual_f32, A_bf16, B_bf16^T, out_dtype=f32) run as one hipBLASLt GEMM (beta = 1)?"""
import os
os.environ.setdefault('TENSILE_STREAMK_DATA_PARALLEL', '1')
import torch
from torch.profiler import profile, ProfilerActivity

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for M, K, N in ((68167, 3072, 256), (68167, 768, 256), (14276, 3072, 256), (118096, 1536, 128)):
    a = torch.randn(M, K, device='cuda').bfloat16()
    b = torch.randn(N, K, device='cuda').bfloat16()
    x = torch.randn(M, N, device='cuda')
    bias = torch.randn(N, device='cuda')
    ref = x + a.float() @ b.float().t()
    y = torch.addmm(x, a, b.t(), out_dtype=torch.float32)
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    yb = torch.addmm(bias, a, b.t(), out_dtype=torch.float32)
    errb = ((yb - (bias + a.float() @ b.float().t())).abs().max() / ref.abs().max()).item()
    t_mm = t(lambda: torch.mm(a, b.t(), out_dtype=torch.float32))
    t_addmm = t(lambda: torch.addmm(x, a, b.t(), out_dtype=torch.float32))
    t_bias = t(lambda: torch.addmm(bias, a, b.t(), out_dtype=torch.float32))
    t_sep = t(lambda: torch.mm(a, b.t(), out_dtype=torch.float32) + x)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        torch.addmm(x, a, b.t(), out_dtype=torch.float32); torch.cuda.synchronize()
    names = [e.key[:40] for e in prof.key_averages()]
    print('M=%d K=%d N=%d err %.1e / %.1e | mm %.1f us, addmm(residual) %.1f us, addmm(bias) %.1f us, mm + add %.1f us | kernels %s'
          % (M, K, N, err, errb, t_mm, t_addmm, t_bias, t_sep, names))
