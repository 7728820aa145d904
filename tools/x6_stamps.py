"""Half-step timeline of hfl_linear_x6 (probe build: HFL_EXTRA_HIPCC_FLAGS=-DHFL_X6_STAMPS python -m hotformerloc_amd.build --force):
s_memtime at every barrier arrival / release of waves 0 (group A) and 4 (group B) of workgroup 0, first tile.
python tools/x6_stamps.py M K N shape"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import ops, _native
m, k, n, shape = (int(v) for v in sys.argv[1:5])
lib = _native.load()
lib.hfl_internal_set_x6_mt.argtypes = [ctypes.c_int]
lib.hfl_internal_set_x6_mt(shape)
x = torch.randn(m, k, device='cuda')
w3 = ops.x6_pack(torch.randn(n, k, device='cuda') * 0.05)
for _ in range(3):
    ops.linear_x6(x, w3)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
lib.hfl_internal_x6_stamps.argtypes = [ctypes.c_void_p]
assert lib.hfl_internal_x6_stamps(buf) == 0
nb = 2 * (2 * ((k + 63) // 64 * 2) + 1)
for g, name in ((0, 'A'), (1, 'B')):
    t = [buf[g * 512 + i] for i in range(min(nb, 512))]
    t0 = buf[0]
    print('group', name, 'barriers (arrive, release) relative to A\'s first, cycles:')
    print('  ' + ' '.join('%d/%d' % (t[2 * i] - t0, t[2 * i + 1] - t0) for i in range(len(t) // 2)))
    print('  work between release and next arrival: ' + ' '.join('%d' % (t[2 * i + 2] - t[2 * i + 1]) for i in range(len(t) // 2 - 1)))
