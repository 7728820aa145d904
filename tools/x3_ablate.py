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
for (M, K, N) in [(68167, 256, 1024), (68167, 1024, 256), (118096, 128, 512)]:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05
    x2 = ops.split2(x); w2 = ops.split2_weight(w)
    for ab, name in ((0, 'full'), (1, 'no in-loop DMA'), (2, 'no mfma'), (4, 'no stores'), (8, 'no lds reads'), (9, 'no DMA, no lds reads'),
                     (6, 'no mfma no stores'), (15, 'nothing but barriers'), (11, 'stores only'), (16, 'nontemporal stores')):
        lib.hfl_set_variant(b'x3_dbg', ab)
        t = timeit(lambda: ops.linear_x3(x2, w2))
        print('M=%d K=%d N=%d %-24s %8.1f us' % (M, K, N, name, t))
    lib.hfl_set_variant(b'x3_dbg', 0)

