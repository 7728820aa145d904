"""One shape of the fused MLP launch, a few launches (for rocprofv3 --pmc surveys): python tools/mlp_fused_one.py [rows] [C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator().manual_seed(1)
x = torch.randn(rows, C, generator=g).cuda()
w1 = (torch.randn(4 * C, C, generator=g) * 0.05).cuda()
w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).cuda()
b1, b2 = torch.zeros(4 * C, device='cuda'), torch.zeros(C, device='cuda')
gamma, beta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
pack = ops.mlp_fused_pack(w1, w2)
out = torch.empty_like(x)
for _ in range(6):
    ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2, out=out)
torch.cuda.synchronize()
