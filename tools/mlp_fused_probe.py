"""Timing of the fused LN -> fc1 -> GELU -> fc2 -> residual launch against the three launches it replaces, per shape of
the bench workload (rows of the pyramid depths of the Wild-Places B = 32 batch).  `python tools/mlp_fused_probe.py`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import _native, ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
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
    dev = 'cuda'
    g = torch.Generator().manual_seed(1)
    for rows, C in ((68167, 256), (14276, 256), (2092, 256), (1902, 256), (118096, 128), (65536, 256)):
        x = torch.randn(rows, C, generator=g).to(dev)
        w1 = (torch.randn(4 * C, C, generator=g) * 0.05).to(dev)
        w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).to(dev)
        b1, b2 = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        pack = ops.mlp_fused_pack(w1, w2)
        w1s, w2s = ops.split2_weight(w1), ops.split2_weight(w2)
        out = torch.empty_like(x)

        def fused():
            ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2, out=out)

        def unfused():
            h2 = ops.layer_norm_split2(x, gamma, beta, 1e-5)
            g2 = ops.linear_x3(h2, w1s, bias=b1, gelu_split_out=True)
            ops.linear_x3(g2, w2s, bias=b2, residual=x, out=out)

        lib = _native.load()
        ts = []
        for groups, sg in ((2, 0), (2, 4), (2, 6), (2, 8), (4, 2), (4, 3), (4, 4), (8, 1), (8, 2)):
            lib.hfl_set_variant(b'mlp_stagger', sg | (groups << 8))
            ts.append(((groups, sg), round(timeit(fused), 1)))
        lib.hfl_set_variant(b'mlp_stagger', 1 | (8 << 8))
        print('   stagger sweep ((groups, naps of ~4000 cycles per group step), us):', ts, flush=True)
        tf, tu = timeit(fused), timeit(unfused)
        flop = 16.0 * rows * C * C * 3
        print('rows %6d C %3d: fused %7.1f us (%6.1f TF/s bf16, %5.2f TB/s alg)   unfused %7.1f us   x%.2f'
              % (rows, C, tf, flop / tf / 1e6, rows * C * 8 / tf / 1e6, tu, tu / tf), flush=True)


if __name__ == '__main__':
    main()
