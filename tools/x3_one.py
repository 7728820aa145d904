"""One shape of the hand-written split GEMM, a few launches (rocprofv3 --pmc surveys):
python tools/x3_one.py [rows] [K] [N] [epi: 0 f32+residual, 1 gelu-split, 2 qkv]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 68167
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = int(sys.argv[3]) if len(sys.argv) > 3 else 768
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 2
g = torch.Generator().manual_seed(1)
x2 = ops.split2(torch.randn(rows, K, generator=g).cuda())
w2 = ops.split2_weight((torch.randn(N, K, generator=g) * 0.05).cuda())
b = torch.zeros(N, device='cuda')
res = torch.randn(rows, N, generator=g).cuda() if epi == 0 else None


def run():
    if epi == 2:
        return ops.linear_x3_qkv(x2, w2, b, 0.36)
    if epi == 1:
        return ops.linear_x3(x2, w2, bias=b, gelu_split_out=True)
    return ops.linear_x3(x2, w2, bias=b, residual=res)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e3
print('rows %d K %d N %d epi %d: %.1f us, %.0f TF/s bf16 (x3), %.2f TB/s algorithmic' % (rows, K, N, epi, t, 6.0 * rows * K * N / t / 1e6,
                                                                                      (rows * K * 4 + rows * N * 4) / t / 1e6))
