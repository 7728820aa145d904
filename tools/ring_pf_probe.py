"""Weight-stream look-ahead of the row-tile kernels (hfl_ln_mlp_fused, hfl_ln_qkv_fused): 2 vs 3 stages in flight ahead of
the consumed one (probe knob 'ring_pf').  Same bits expected either way (the arithmetic does not change); timing per shape
of the bench workload.  `python tools/ring_pf_probe.py`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import _native, ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _native.load()
    g = torch.Generator().manual_seed(1)
    dev = 'cuda'
    for rows, C in ((68167, 256), (66775, 256), (65536, 256), (32768, 256), (40000, 256), (14276, 256), (118096, 128), (131072, 128)):
        x = (torch.randn(rows, C, generator=g) * 1.5 + 0.3).to(dev)
        w1 = (torch.randn(4 * C, C, generator=g) * 0.05).to(dev)
        w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).to(dev)
        wq = (torch.randn(3 * C, C, generator=g) * 0.06).to(dev)
        b1, b2 = (torch.randn(4 * C, generator=g) * 0.1).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        bq = (torch.randn(3 * C, generator=g) * 0.1).to(dev)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        mpack, qpack = ops.mlp_fused_pack(w1, w2), ops.qkv_fused_pack(wq)
        out = torch.empty_like(x)
        qout = torch.empty((rows, 3 * C), dtype=torch.float32, device=dev)
        res = {}
        lib.hfl_set_variant(b'tail_split', 0)
        lib.hfl_set_variant(b'ring_pf', 3)
        t_nosplit = (timeit(lambda: ops.ln_mlp_fused(x, gamma, beta, 1e-5, mpack, b1, b2, out=out)),
                     timeit(lambda: ops.ln_qkv_fused(x, gamma, beta, 1e-5, qpack, bq, 0.36, out=qout)))
        lib.hfl_set_variant(b'tail_split', 1)
        print('rows %6d C %3d | pf3 WITHOUT tail split: mlp %.1f us  qkv %.1f us' % (rows, C, t_nosplit[0], t_nosplit[1]))
        for pf in (2, 3, 2, 3):
            lib.hfl_set_variant(b'ring_pf', pf)
            tm = timeit(lambda: ops.ln_mlp_fused(x, gamma, beta, 1e-5, mpack, b1, b2, out=out))
            tq = timeit(lambda: ops.ln_qkv_fused(x, gamma, beta, 1e-5, qpack, bq, 0.36, out=qout))
            res.setdefault(pf, []).append((tm, tq, out.clone(), qout.clone()))
        same_m = torch.equal(res[2][0][2], res[3][0][2])
        same_q = torch.equal(res[2][0][3].view(torch.int32), res[3][0][3].view(torch.int32))
        flop_m, flop_q = 16.0 * rows * C * C * 3, 6.0 * rows * C * C * 3
        print('rows %6d C %3d | mlp pf2 %s us  pf3 %s us (%.0f TF/s bf16) bits equal %s | qkv pf2 %s us  pf3 %s us (%.0f TF/s) bits equal %s'
              % (rows, C, ['%.1f' % r[0] for r in res[2]], ['%.1f' % r[0] for r in res[3]],
                 flop_m / min(r[0] for r in res[3]) / 1e6, same_m,
                 ['%.1f' % r[1] for r in res[2]], ['%.1f' % r[1] for r in res[3]],
                 flop_q / min(r[1] for r in res[3]) / 1e6, same_q), flush=True)
    lib.hfl_set_variant(b'reset', 0)


if __name__ == '__main__':
    main()
