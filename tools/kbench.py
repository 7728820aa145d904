"""Kernel micro-benchmarks at bench shapes (Wild-Places, B=32): A/B of kernel variants in ONE
process, interleaved rounds, HIP-event timing (cdna guide rule 24)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import build_batch_octree, load_config, ops, _native, synthetic as syn
from hotformerloc_amd.plan import WindowPlan

def timeit(fn, rounds=12, inner=10):
    """median / min microseconds per call; `inner` calls are queued back to back between the two
    events so that host launch overhead (tens of us per ctypes call) does not count."""
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    ts.sort()
    return ts[len(ts) // 2], ts[0]

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'wild-places'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    params, depth = load_config(cfg)
    clouds = syn.make_clouds(2, B, 4096, params.coordinates)
    octree = build_batch_octree(clouds, depth, 2, 'cuda')
    md = depth - 2
    plan = WindowPlan(octree, params.patch_size, params.dilation, md, md - 3, 3, 1, params.ADaPE_mode)
    lib = _native.load()
    K = params.patch_size
    g = torch.Generator(device='cuda').manual_seed(0)
    print('config', cfg, 'B', B, 'tokens', plan.n_tokens, 'windows', plan.n_windows)
    for d, H, G, dil in ((md, 8, 0, 1), (md, 8, 0, params.dilation), (md - 1, 16, 1, 1), (md - 2, 16, 1, 1), (md - 3, 16, 1, 1)):
        C = H * 16
        nt, W = plan.n_tokens[d], plan.n_windows[d]
        rows = nt + (W if G else 0)
        qkv = torch.randn(rows, 3 * C, device='cuda', generator=g)
        bnd = int(0.8 * K * dil ** 0.5)
        table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.1
        nbytes = rows * C * 16
        flops = 4 * (K + G) ** 2 * C * (-(-nt // K))
        res = {}
        for name, var, wgs, dbg in (('v2', 2, 16, 0), ('v4/x1', 4, 1, 0), ('v4/x2', 4, 2, 0), ('v4/x3', 4, 3, 0)):
            lib.hfl_set_variant(b'window_attention', var)
            lib.hfl_set_variant(b'window_v4_wgs_per_cu', wgs)
            lib.hfl_set_variant(b'window_debug', dbg)
            for split in (True,):
                f = lambda: ops.window_attention(qkv, plan.meta[d], table, nt, W, K, dil, G, H, B, rt_row0=nt, depth=d, out_split=split)
                med, mn = timeit(f)
                print('window_attn d=%d H=%d G=%d D=%d split=%d %-18s med %7.1f us  min %7.1f us  %6.0f GB/s  %5.1f TF/s' %
                      (d, H, G, dil, split, name, med, mn, (nbytes + (rows * C * 2 if split else 0)) / med / 1e3, flops / med / 1e6))
        lib.hfl_set_variant(b'window_debug', 0)
    lib.hfl_set_variant(b'window_attention', 4)
    lib.hfl_set_variant(b'window_heads_per_wg', 4)
    lib.hfl_set_variant(b'window_v4_wgs_per_cu', 4)
    lib.hfl_set_variant(b'window_v2_wgs_per_cu', 16)
    # CPE
    for d, C in ((md, 128), (md - 1, 256), (md - 2, 256)):
        n = plan.n_tokens[d]
        x = torch.randn(n, C, device='cuda', generator=g)
        w = torch.randn(27, 1, C, device='cuda', generator=g)
        gm = torch.ones(C, device='cuda'); bt = torch.zeros(C, device='cuda')
        neigh = plan.neigh(d)
        nb = n * C * 8 + n * 27 * 4
        ref = None
        for var, wgs in ((0, 0), (1, 3)):
            lib.hfl_set_variant(b'cpe_variant', var)
            if wgs:
                lib.hfl_set_variant(b'cpe_lds_wgs_per_cu', wgs)
            f = lambda: ops.cpe_forward(x, w, gm, bt, neigh, True)
            med, mn = timeit(f)
            o = f()
            if ref is None:
                ref = o
            else:
                assert (o - ref).abs().max().item() < 2e-5, (o - ref).abs().max().item()
            print('cpe d=%d C=%d n=%d variant=%d wgs/cu=%d med %7.1f us  min %7.1f us  %6.0f GB/s' % (d, C, n, var, wgs, med, mn, nb / med / 1e3))
        lib.hfl_set_variant(b'cpe_variant', 0)
        lib.hfl_set_variant(b'cpe_lds_wgs_per_cu', 3)
        med, mn = timeit(lambda: torch.nn.functional.layer_norm(x, (C,), gm, bt))
        print('   torch LN same shape   med %7.1f us  %6.0f GB/s' % (med, n * C * 8 / med / 1e3))
        y = torch.randn_like(x)
        med, mn = timeit(lambda: x + y)
        print('   torch add same shape  med %7.1f us  %6.0f GB/s' % (med, n * C * 12 / med / 1e3))

if __name__ == '__main__':
    main()
