"""Timing ablations of the fused MLP launch (probe knob 'mlp_dbg'; the ablated launches compute wrong results):
what a pass is made of.  `python tools/mlp_ablate.py [rows]`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import _native, ops  # noqa: E402
from ring_pf_probe import timeit  # noqa: E402

lib = _native.load()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
C = 256
g = torch.Generator().manual_seed(1)
x = (torch.randn(rows, C, generator=g) * 1.5 + 0.3).cuda()
w1 = (torch.randn(4 * C, C, generator=g) * 0.05).cuda()
w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).cuda()
b1, b2 = (torch.randn(4 * C, generator=g) * 0.1).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
pack = ops.mlp_fused_pack(w1, w2)
out = torch.empty_like(x)
names = {0: 'as shipped', 1: 'GELU -> identity', 3: 'no bias / GELU / split', 4: 'no ring refill', 7: 'no GELU / split, no refill',
         8: 'no stage barrier', 15: 'none of them'}
for rep in range(2):
    for d in (0, 1, 3, 4, 7, 8, 15):
        lib.hfl_set_variant(b'mlp_dbg', d)
        t = timeit(lambda: ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2, out=out))
        print('rows %d  %-28s %.1f us' % (rows, names[d], t), flush=True)
lib.hfl_set_variant(b'reset', 0)
