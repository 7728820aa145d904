"""(second mode, `python tools/mlp_waves_probe.py lag`: the second wave of every SIMD one stage behind the first, knob
'mlp_lag', against the lock-step kernel with two and three stages of weight stream in flight.)
hfl_ln_mlp_fused with 8 waves per workgroup (two per SIMD, one 16-row tile each at C = 256) against 4 waves (one per SIMD,
512 registers, two tiles each): same bits expected, timing per shape of the bench workload.  `python tools/mlp_waves_probe.py`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import _native, ops  # noqa: E402
from ring_pf_probe import timeit  # noqa: E402


def lag_mode():
    lib = _native.load()
    g = torch.Generator().manual_seed(1)
    dev = 'cuda'
    for rows, C in ((68167, 256), (65536, 256), (32768, 256), (14276, 256), (2092, 256), (1902, 256), (118096, 128), (131072, 128), (777, 128)):
        x = (torch.randn(rows, C, generator=g) * 1.5 + 0.3).to(dev)
        w1 = (torch.randn(4 * C, C, generator=g) * 0.05).to(dev)
        w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).to(dev)
        b1, b2 = (torch.randn(4 * C, generator=g) * 0.1).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        mpack = ops.mlp_fused_pack(w1, w2)
        out = torch.empty_like(x)
        res = {}
        for mode in ('pf3', 'pf2', 'lag', 'pf3', 'pf2', 'lag'):
            lib.hfl_set_variant(b'ring_pf', 2 if mode == 'pf2' else 3)
            lib.hfl_set_variant(b'mlp_lag', 1 if mode == 'lag' else 0)
            tm = timeit(lambda: ops.ln_mlp_fused(x, gamma, beta, 1e-5, mpack, b1, b2, out=out))
            res.setdefault(mode, []).append((tm, out.clone()))
        same = torch.equal(res['pf3'][0][1].view(torch.int32), res['lag'][0][1].view(torch.int32))
        flop = 16.0 * rows * C * C * 3
        print('rows %6d C %3d | lock-step, 3 ahead %s us  2 ahead %s us | one stage apart %s us (%.0f TF/s bf16)  bits equal %s'
              % (rows, C, ['%.1f' % r[0] for r in res['pf3']], ['%.1f' % r[0] for r in res['pf2']],
                 ['%.1f' % r[0] for r in res['lag']], flop / min(r[0] for r in res['lag']) / 1e6, same), flush=True)
    lib.hfl_set_variant(b'reset', 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'lag':
        return lag_mode()
    lib = _native.load()
    g = torch.Generator().manual_seed(1)
    dev = 'cuda'
    for rows, C in ((68167, 256), (65536, 256), (32768, 256), (14276, 256), (2092, 256), (118096, 128), (131072, 128), (777, 128)):
        x = (torch.randn(rows, C, generator=g) * 1.5 + 0.3).to(dev)
        w1 = (torch.randn(4 * C, C, generator=g) * 0.05).to(dev)
        w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).to(dev)
        b1, b2 = (torch.randn(4 * C, generator=g) * 0.1).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
        mpack = ops.mlp_fused_pack(w1, w2)
        wq = (torch.randn(3 * C, C, generator=g) * 0.06).to(dev)
        bq = (torch.randn(3 * C, generator=g) * 0.1).to(dev)
        qpack = ops.qkv_fused_pack(wq)
        out = torch.empty_like(x)
        qout = torch.empty((rows, 3 * C), dtype=torch.float32, device=dev)
        res = {}
        for nw in (8, 4, 8, 4):
            lib.hfl_set_variant(b'mlp_waves', nw)
            lib.hfl_set_variant(b'qkv_waves', nw)
            tm = timeit(lambda: ops.ln_mlp_fused(x, gamma, beta, 1e-5, mpack, b1, b2, out=out))
            tq = timeit(lambda: ops.ln_qkv_fused(x, gamma, beta, 1e-5, qpack, bq, 0.36, out=qout)) if rows >= 4096 else 0.0
            res.setdefault(nw, []).append((tm, out.clone(), tq, qout.clone()))
        same = torch.equal(res[8][0][1].view(torch.int32), res[4][0][1].view(torch.int32))
        same_q = torch.equal(res[8][0][3].view(torch.int32), res[4][0][3].view(torch.int32))
        flop = 16.0 * rows * C * C * 3
        print('rows %6d C %3d | mlp 8 waves %s us (%.0f TF/s bf16)   4 waves %s us (%.0f TF/s)   bits equal %s | qkv 8 waves %s  4 waves %s  bits equal %s'
              % (rows, C, ['%.1f' % r[0] for r in res[8]], flop / min(r[0] for r in res[8]) / 1e6,
                 ['%.1f' % r[0] for r in res[4]], flop / min(r[0] for r in res[4]) / 1e6, same,
                 ['%.1f' % r[2] for r in res[8]], ['%.1f' % r[2] for r in res[4]], same_q), flush=True)
    lib.hfl_set_variant(b'reset', 0)


if __name__ == '__main__':
    main()
