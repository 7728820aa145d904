"""hfl_dwconv_weight_backward at the training step's shapes: error against an fp64 contraction and time per launch for the
gather batch sizes of the kernel (csrc/dwconv.hip: dwconv_wgrad_partial<.., B>)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn, _native
from hotformerloc_amd.plan import WindowPlan

lib = _native.load()
lib.hfl_internal_set_wgrad_batch.argtypes = [ctypes.c_int]
params, depth = load_config('wild-places')
octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
plan = WindowPlan(octree, 48, 4, 5, 2, 3, 1, None)
g = torch.Generator(device='cuda').manual_seed(0)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for d, C in ((4, 256), (5, 128), (3, 256)):
    neigh = plan.neigh(d)
    n = neigh.shape[0]
    x = torch.randn(n, C, device='cuda', generator=g)
    dy = torch.randn(n, C, device='cuda', generator=g)
    ref = torch.zeros(27, C, dtype=torch.float64, device='cuda')
    for k in range(27):
        idx = neigh[:, k].long()
        ok = idx >= 0
        ref[k] = (x[idx[ok]].double() * dy[ok].double()).sum(0)
    line = 'depth %d  rows %6d  C %3d  live taps %.1f |' % (d, n, C, float((neigh >= 0).float().sum(1).mean()))
    for b in (9, 6, 3):
        lib.hfl_internal_set_wgrad_batch(b)
        dw = ops.dwconv_weight_backward(dy, x, neigh).view(27, C)
        err = ((dw.double() - ref).norm() / ref.norm()).item()
        t = timeit(lambda: ops.dwconv_weight_backward(dy, x, neigh))
        line += '  B=%d: %.1f us (err %.1e)' % (b, t, err)
    lib.hfl_internal_set_wgrad_batch(9)
    w = torch.randn(27, 1, C, device='cuda', generator=g)
    gm = torch.ones(C, device='cuda'); bt = torch.zeros(C, device='cuda')
    tf = timeit(lambda: ops.cpe_forward(x, w, gm, bt, neigh, True))
    print(line + '  | CPE forward %.1f us' % tf)
