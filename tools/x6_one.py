"""One shape of hfl_linear_x6 for counter surveys: python tools/x6_one.py M K N [mt] [gelu]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import ops, _native
m, k, n = (int(v) for v in sys.argv[1:4])
mt = int(sys.argv[4]) if len(sys.argv) > 4 else 0
gelu = len(sys.argv) > 5 and sys.argv[5] == '1'
lib = _native.load()
lib.hfl_internal_set_x6_mt.argtypes = [ctypes.c_int]
lib.hfl_internal_set_x6_mt(mt)
torch.manual_seed(0)
x = torch.randn(m, k, device='cuda')
w3 = ops.x6_pack(torch.randn(n, k, device='cuda') * 0.05)
b = torch.randn(n, device='cuda')
for _ in range(10):
    ops.linear_x6(x, w3, bias=b, gelu=gelu)
torch.cuda.synchronize()
