"""When does hipBLASLt read TENSILE_STREAMK_DATA_PARALLEL?  usage: sk_env_probe.py early|late|never"""
import os, sys
mode = sys.argv[1]
if mode == 'early':
    os.environ['TENSILE_STREAMK_DATA_PARALLEL'] = '1'
import torch
if mode == 'late':
    os.environ['TENSILE_STREAMK_DATA_PARALLEL'] = '1'
from torch.profiler import profile, ProfilerActivity
M, K, N = 68167, 768, 1024
a = torch.randn(M, K, device='cuda').bfloat16()
b = torch.randn(N, K, device='cuda').bfloat16()
for _ in range(3):
    torch.mm(a, b.t(), out_dtype=torch.float32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    torch.mm(a, b.t(), out_dtype=torch.float32)
e1.record(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    torch.mm(a, b.t(), out_dtype=torch.float32); torch.cuda.synchronize()
names = [e.key for e in prof.key_averages() if 'Cijk' in e.key]
sk = [n[n.find('_SK'):n.find('_SK') + 4] for n in names]
mt = [n[n.find('_MT'):n.find('_MT') + 14] for n in names]
print(mode, '%.1f us/GEMM' % (e0.elapsed_time(e1) * 1e3 / 20), mt, sk)
