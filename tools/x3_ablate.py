"""A/B of the x3 GEMM's row-tile height (128 vs 256 rows per workgroup) on the model's shapes, all three epilogues."""
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
for (M, K, N, tag) in [(68167, 256, 768, 'qkv d4'), (68167, 256, 256, 'proj d4'), (68167, 256, 1024, 'fc1 d4'),
                       (68167, 1024, 256, 'fc2 d4'), (118096, 128, 384, 'qkv d5'), (118096, 128, 128, 'proj d5'),
                       (118096, 128, 512, 'fc1 d5'), (118096, 512, 128, 'fc2 d5'), (14276, 256, 1024, 'fc1 d3'),
                       (14276, 1024, 256, 'fc2 d3'), (14276, 256, 768, 'qkv d3')]:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05; b = torch.randn(N, device='cuda')
    res = torch.randn(M, N, device='cuda')
    x2 = ops.split2(x); w2 = ops.split2_weight(w)
    out = {}
    for mt in (4, 8):
        lib.hfl_set_variant(b'x3_dbg', 0x200 | mt)
        y = ops.linear_x3(x2, w2, bias=b, residual=res)
        g = ops.linear_x3(x2, w2, bias=b, gelu_split_out=True)
        out[mt] = (y, g, timeit(lambda: ops.linear_x3(x2, w2, bias=b, residual=res)),
                   timeit(lambda: ops.linear_x3(x2, w2, bias=b, gelu_split_out=True)))
    lib.hfl_set_variant(b'x3_dbg', 0x200)
    same = torch.equal(out[4][0], out[8][0]) and torch.equal(out[4][1], out[8][1])
    print('%-8s M=%6d K=%4d N=%4d | +res: 128-row %7.1f us, 256-row %7.1f us | gelu-epi: %7.1f / %7.1f us | bit-equal %s'
          % (tag, M, K, N, out[4][2], out[8][2], out[4][3], out[8][3], same))
