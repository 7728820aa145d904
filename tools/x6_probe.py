"""hfl_linear_x6 (csrc/gemm_x6.hip) against fp64 and against the fp32 library GEMM: error and time at the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from hotformerloc_amd import ops, _native
lib = _native.load()
lib.hfl_internal_set_x6_mt.argtypes = [__import__('ctypes').c_int]

torch.manual_seed(0)
dev = 'cuda'


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item()


shapes = [(68167, 256, 768, 'qkv d4'), (68167, 256, 256, 'proj d4'), (68167, 256, 1024, 'fc1 d4'), (68167, 1024, 256, 'fc2 d4'),
          (118096, 128, 384, 'qkv d5'), (118096, 128, 512, 'fc1 d5'), (118096, 512, 128, 'fc2 d5'),
          (14310, 256, 1024, 'fc1 d3'), (2100, 256, 1024, 'fc1 d2'), (1733, 1024, 256, 'fc2 rt'), (300, 256, 256, 'tiny'),
          (257, 32, 128, 'edge'), (5000, 96, 128, 'k96')]
for m, k, n, name in shapes:
    x = torch.randn(m, k, device=dev) * 1.3
    w = torch.randn(n, k, device=dev) * 0.05
    b = torch.randn(n, device=dev) * 0.1
    r = torch.randn(m, n, device=dev)
    w3 = ops.x6_pack(w)
    ref = x.double() @ w.double().t() + b.double()
    y6 = ops.linear_x6(x, w3, bias=b)
    y32 = F.linear(x, w, b)
    e6, e32 = rel(y6, ref), rel(y32, ref)
    yr = ops.linear_x6(x, w3, bias=b, residual=r)
    er = rel(yr, ref + r.double())
    yg = ops.linear_x6(x, w3, bias=b, gelu=True)
    eg = rel(yg, F.gelu(ref))
    # in place: residual aliases out
    r2 = r.clone()
    ops.linear_x6(x, w3, bias=b, residual=r2, out=r2)
    ei = rel(r2, ref + r.double())
    t6 = timeit(lambda: ops.linear_x6(x, w3, bias=b))
    tmt = []
    for mt in (2, 4, 12, 14):
        lib.hfl_internal_set_x6_mt(mt)
        tmt.append(timeit(lambda: ops.linear_x6(x, w3, bias=b, residual=r)))
    lib.hfl_internal_set_x6_mt(0)
    tres = timeit(lambda: ops.linear_x6(x, w3, bias=b, residual=r))
    t6g = timeit(lambda: ops.linear_x6(x, w3, bias=b, gelu=True))
    t32 = timeit(lambda: F.linear(x, w, b))
    x2 = ops.split2(x) if k % 32 == 0 else None
    w2 = ops.split2_weight(w)
    t3 = timeit(lambda: ops.linear_x3(x2, w2, bias=b))
    fl = 2.0 * m * k * n
    print('%-8s M=%6d K=%4d N=%4d | err vs fp64: x6 %.2e  fp32 lib %.2e | +res %.2e  gelu %.2e  inplace %.2e | x6 %7.1f us (%5.1f TF, %4.2f of 2.5 PF issued)  x6+gelu %7.1f  fp32 lib %7.1f us (%5.1f TF)  x3 %7.1f us'
          % (name, m, k, n, e6, e32, er, eg, ei, t6, fl / t6 / 1e6, 6 * fl / t6 / 1e6 / 2500.0, t6g, t32, fl / t32 / 1e6, t3))
    print('         +residual: auto %.1f us; forced tile 128/256 x 128, 64/128 x 256: %s us' % (tres, ' '.join('%.1f' % t for t in tmt)))
