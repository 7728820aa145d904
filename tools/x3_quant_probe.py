"""Does the workgroup count quantise the x3 GEMM's time?  768 slots (3 per CU): time vs number of 128-row tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K, N in ((256, 256), (1024, 256), (256, 768)):
    w = torch.randn(N, K, device='cuda') * 0.05; b = torch.randn(N, device='cuda')
    w2 = ops.split2_weight(w)
    for wgs in (384, 576, 768, 800, 960, 1066, 1152, 1344, 1536, 1600, 2304):
        M = wgs * 128 // (N // 128)
        x2 = ops.split2(torch.randn(M, K, device='cuda')); res = torch.randn(M, N, device='cuda')
        t = timeit(lambda: ops.linear_x3(x2, w2, bias=b, residual=res))
        print('K=%4d N=%4d  %5d workgroups (M=%6d): %7.1f us  (%.3f us per workgroup-slot round of 768)' % (K, N, wgs, M, t, t / -(-wgs // 768)))
