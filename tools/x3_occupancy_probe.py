"""x3 GEMM time against the number of co-resident workgroups per CU (extra dynamic LDS cuts the occupancy 3 -> 2 -> 1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops, _native
lib = _native.load()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, K, N, tag) in [(68167, 256, 1024, 'fc1 d4'), (68167, 256, 768, 'qkv d4'), (68167, 1024, 256, 'fc2 d4'), (118096, 128, 512, 'fc1 d5')]:
    x2 = ops.split2(torch.randn(M, K, device='cuda')); w2 = ops.split2_weight(torch.randn(N, K, device='cuda') * 0.05); b = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda')
    out = []
    for extra, wgs in ((0, 3), (24, 2), (60, 1)):
        lib.hfl_set_variant(b'x3_dbg', extra)
        out.append('%d per CU: gelu-epi %.1f us, +res %.1f us' % (wgs, timeit(lambda: ops.linear_x3(x2, w2, bias=b, gelu_split_out=True)),
                                                              timeit(lambda: ops.linear_x3(x2, w2, bias=b, residual=res))))
    lib.hfl_set_variant(b'x3_dbg', 0)
    print(tag, ' | '.join(out))
