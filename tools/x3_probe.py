"""Probe: hand-written split2 GEMM (hfl_linear_x3) vs the hipBLASLt K-concatenated route, on the model's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops
dev = 'cuda'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
torch.manual_seed(0)
for (M, K, N, tag) in [(68167, 256, 768, 'qkv d4'), (68167, 256, 256, 'proj d4'), (68167, 256, 1024, 'fc1 d4'),
                       (68167, 1024, 256, 'fc2 d4'), (118096, 128, 384, 'qkv d5'), (118096, 128, 128, 'proj d5'),
                       (118096, 128, 512, 'fc1 d5'), (118096, 512, 128, 'fc2 d5'), (14276, 256, 1024, 'fc1 d3'),
                       (2100, 256, 1024, 'fc1 d2'), (130, 256, 768, 'tiny')]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    x2 = ops.split2(x); w2 = ops.split2_weight(w)
    a3 = ops.split3(x); w3 = ops.split_weight(w)
    y = ops.linear_x3(x2, w2, bias=b)
    err = ((y.double() - ref).norm() / ref.norm()).item()
    yr = ops.linear_x3(x2, w2, bias=b, residual=res)
    err_r = ((yr.double() - (ref + res.double())).norm() / (ref + res.double()).norm()).item()
    g2 = ops.linear_x3(x2, w2, bias=b, gelu_split_out=True)
    gref = torch.nn.functional.gelu(ref)
    gg = g2.view(M, N // 32, 2, 32).float()
    gval = (gg[:, :, 0] + gg[:, :, 1]).reshape(M, N)
    err_g = ((gval.double() - gref).norm() / gref.norm()).item()
    y3 = ops.gemm_bf16(a3, w3, bias=b)
    err3 = ((y3.double() - ref).norm() / ref.norm()).item()
    t_x3 = timeit(lambda: ops.linear_x3(x2, w2, bias=b))
    t_x3r = timeit(lambda: ops.linear_x3(x2, w2, bias=b, residual=res))
    t_x3g = timeit(lambda: ops.linear_x3(x2, w2, bias=b, gelu_split_out=True))
    t_lt = timeit(lambda: ops.gemm_bf16(a3, w3, bias=b))
    t_ltr = timeit(lambda: ops.gemm_bf16(a3, w3, bias=b, residual=res))
    hid = torch.randn(M, N, device=dev)
    t_gelu = timeit(lambda: ops.bias_gelu_split3(hid, b))
    fl = 2.0 * M * K * N
    print('%-8s M=%6d K=%4d N=%4d | x3 %.3f ms (%4.0f TF eff, err %.1e) +res %.3f (err %.1e) gelu-epi %.3f (err %.1e) | '
          'hipBLASLt %.3f ms (%4.0f TF, err %.1e) +res %.3f  | separate gelu+split pass %.3f ms'
          % (tag, M, K, N, t_x3, fl / t_x3 / 1e9, err, t_x3r, err_r, t_x3g, err_g, t_lt, fl / t_lt / 1e9, err3, t_ltr, t_gelu))
