"""A few launches of hfl_linear_x3 at one shape (for rocprofv3 counter passes): tools/x3_only.py M K N [gelu]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import ops
M, K, N = (int(a) for a in sys.argv[1:4])
gelu = len(sys.argv) > 4 and sys.argv[4] == 'gelu'
x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05; b = torch.randn(N, device='cuda')
x2 = ops.split2(x); w2 = ops.split2_weight(w)
for _ in range(6):
    ops.linear_x3(x2, w2, bias=b, gelu_split_out=gelu)
torch.cuda.synchronize()
