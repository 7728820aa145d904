"""Phase timeline of ONE wave of the fp16 window-attention kernel (window_debug bit 3: s_memtime stamps at the top of the window
loop, after the barrier, after the fragment setup / prefetch issue and after every query tile), depth-4 bench shape.
HFL_KNOBS=key=value,... sets probe knobs (hfl_set_variant).  Needs a library built with the stamps compiled in:
`HFL_EXTRA_HIPCC_FLAGS=-DHFL_ATT_TRACE=1 python -m hotformerloc_amd.build` (touch csrc/attention.hip first)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import _native, build_batch_octree, load_config, ops, synthetic as syn  # noqa: E402
from hotformerloc_amd.plan import WindowPlan  # noqa: E402

lib = _native.load()
for kv in os.environ.get('HFL_KNOBS', '').split(','):
    if '=' in kv:
        lib.hfl_set_variant(kv.split('=')[0].encode(), int(kv.split('=')[1]))
d = int(sys.argv[1]) if len(sys.argv) > 1 else 4
params, depth = load_config('wild-places')
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
plan = WindowPlan(octree, 48, 4, 5, 2, 3, 1, None)
g = torch.Generator(device='cuda').manual_seed(0)
H, G, C = (8, 0, 128) if d == 5 else (16, 1, 256)
nt, W = plan.n_tokens[d], plan.n_windows[d]
rows = nt + W * G
x = torch.randn(rows, C, device='cuda', generator=g)
w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.06
b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
qkv = ops.linear_x3_qkv(ops.split2(x), ops.split2_weight(w), b, 16 ** -0.5 * 1.4426950408889634)
table = torch.randn(3 * 77, H, device='cuda', generator=g) * 0.1


def run():
    return ops.window_attention(qkv, plan.meta[d], table, nt, W, 48, 1, G, H, 32, rt_row0=nt, depth=d, out_split=2, qkv_f16=True)


for _ in range(3):
    run()
lib.hfl_set_variant(b'window_debug', 8)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib.hfl_internal_read_att_trace.restype = ctypes.c_int
lib.hfl_internal_read_att_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.hfl_internal_read_att_trace(buf, 128) == 0
t = [[buf[i * 8 + j] for j in range(8)] for i in range(16)]
print('launch %.1f us; stamps in s_memtime ticks relative to the wave\'s first: [top, barrier, setup, tile0..tile3]' % (e0.elapsed_time(e1) * 1e3))
t0 = t[0][0]
last = max(i for i in range(16) if t[i][0])
if last > 0:
    print('s_memtime ticks per us (against s_memrealtime at 100 MHz): %.0f' % ((t[last][0] - t[0][0]) / ((t[last][7] - t[0][7]) / 100.0)))
nq = 4 if G else 3
for i in range(16):
    if t[i][0] == 0 or (i and t[i][0] < t[i - 1][0]):
        break
    row = [t[i][j] - t0 for j in range(3 + nq)]
    print('window %2d: ' % i + ' '.join('%7d' % v for v in row) + '   | deltas ' + ' '.join('%6d' % (row[j + 1] - row[j]) for j in range(2 + nq)))

import numpy as np  # noqa: E402
wg = (ctypes.c_ulonglong * 8192)()
lib.hfl_internal_read_att_wg.restype = ctypes.c_int
lib.hfl_internal_read_att_wg.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.hfl_internal_read_att_wg(wg, 8192) == 0
a = np.array(list(wg), dtype=np.int64).reshape(-1, 2)
a = a[a[:, 0] > 0]
st = (a[:, 0] - a[:, 0].min()) / 100.0
en = (a[:, 1] - a[:, 0].min()) / 100.0
print('%d workgroups: start (us after the first) min %.1f median %.1f p90 %.1f max %.1f | end min %.1f median %.1f p90 %.1f max %.1f | '
      'lifetime min %.1f median %.1f max %.1f' % (len(a), st.min(), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en),
                                                   np.percentile(en, 90), en.max(), (en - st).min(), np.median(en - st), (en - st).max()))
hist = np.histogram(st, bins=8)
print('start histogram:', list(zip(np.round(hist[1][:-1], 1).tolist(), hist[0].tolist())))
hist = np.histogram(en, bins=8)
print('end histogram:  ', list(zip(np.round(hist[1][:-1], 1).tolist(), hist[0].tolist())))
life = (a[:, 1] - a[:, 0]) / 100.0
n = len(a)
print('lifetime by (workgroup id % 8) [XCD under round-robin dispatch]:', [round(float(life[i::8].mean()), 1) for i in range(8)],
      ' spread inside one class: min %.1f max %.1f' % (life[0::8].min(), life[0::8].max()))
gx = n // 4 if n % 4 == 0 else n
print('lifetime by head group (blockIdx.y):', [round(float(life[i * gx:(i + 1) * gx].mean()), 1) for i in range(n // gx)])
cu = life[:gx].reshape(-1, 8)[:, 0]
print('XCD-0 workgroups of head group 0, in dispatch order:', np.round(cu, 1).tolist())
