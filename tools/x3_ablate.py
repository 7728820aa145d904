"""A/B of the x3 GEMM tile variants on the model's shapes: 128 x 128 (default), 256 x 128 single-stage (knob 0x200 | 8), and the
wide 128 x 256 two-stage 8-wave kernel (knob 0x400 | 1)."""
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
    for name, knobs in (('128x128', (0x200, 0x400)), ('256x128', (0x200 | 8, 0x400)), ('wide 128x256', (0x200, 0x400 | 1))):
        for kb in knobs:
            lib.hfl_set_variant(b'x3_dbg', kb)
        y = ops.linear_x3(x2, w2, bias=b, residual=res)
        g = ops.linear_x3(x2, w2, bias=b, gelu_split_out=True)
        out[name] = (y, g, timeit(lambda: ops.linear_x3(x2, w2, bias=b, residual=res)),
                     timeit(lambda: ops.linear_x3(x2, w2, bias=b, gelu_split_out=True)))
    lib.hfl_set_variant(b'x3_dbg', 0x200)
    lib.hfl_set_variant(b'x3_dbg', 0x400)
    same = all(torch.equal(out['128x128'][0], o[0]) and torch.equal(out['128x128'][1], o[1]) for o in out.values())
    print('%-8s M=%6d K=%4d N=%4d | +res: %s | gelu-epi: %s | bit-equal %s'
          % (tag, M, K, N, ' / '.join('%s %6.1f us' % (k, v[2]) for k, v in out.items()),
             ' / '.join('%6.1f' % v[3] for v in out.values()), same))
