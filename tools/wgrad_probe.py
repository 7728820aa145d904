"""hfl_wgrad_x3 (split2 operands, three-term bf16 MFMA) vs the fp32 GEMM autograd would run, on the model's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, tag) in [(68167, 768, 256, 'qkv d4'), (68167, 256, 256, 'proj d4'), (68167, 1024, 256, 'fc1 d4'),
                       (68167, 256, 1024, 'fc2 d4'), (118096, 384, 128, 'qkv d5'), (118096, 512, 128, 'fc1 d5'),
                       (118096, 128, 512, 'fc2 d5'), (300000, 1024, 256, 'fc1 d4 cs'), (300000, 256, 1024, 'fc2 d4 cs'),
                       (14276, 1024, 256, 'fc1 d3')]:
    dy = torch.randn(M, N, device='cuda'); x = torch.randn(M, K, device='cuda')
    dys, xs = ops.split2(dy), ops.split2(x)
    ref = dy.double().t() @ x.double()
    dw, db = ops.wgrad_x3(dys, xs, with_bias=True)
    err = ((dw.double() - ref).norm() / ref.norm()).item()
    e32 = ((torch.mm(dy.t(), x).double() - ref).norm() / ref.norm()).item()
    t3 = timeit(lambda: ops.wgrad_x3(dys, xs, with_bias=True))
    t32 = timeit(lambda: (torch.mm(dy.t(), x), dy.sum(0)))
    ts = timeit(lambda: ops.split2(dy))
    fl = 2.0 * M * N * K
    print('%-10s M=%6d N=%4d K=%4d | x3 %8.1f us (%4.0f TF eff, %.2f TB/s operands, err %.1e) | fp32 mm + sum %8.1f us (%4.0f TF, err %.1e) | split2(dy) %6.1f us'
          % (tag, M, N, K, t3, fl / t3 / 1e6, M * (N + K) * 4 / t3 / 1e6, err, t32, fl / t32 / 1e6, e32, ts))
