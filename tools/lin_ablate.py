import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops, _native
lib = _native.load()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, K, N) in [(68167, 256, 768), (68167, 1024, 256), (68167, 256, 1024)]:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05
    wh, wl = ops.split_weight_pair(w)
    for ab, name in ((0, 'full'), (1, 'no global loads'), (2, 'no lds stores'), (3, 'no mfma'), (4, 'no epilogue store')):
        lib.hfl_set_variant(b'linear_ablate', ab)
        t = timeit(lambda: ops.linear_bf16x3(x, wh, wl))
        print('M=%d K=%d N=%d %-18s %8.1f us' % (M, K, N, name, t))
    lib.hfl_set_variant(b'linear_ablate', 0)
