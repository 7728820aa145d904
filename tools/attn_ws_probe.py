"""hfl_attn_ws_fwd (LayerNorm -> qkv -> window attention of a relay-token block in one launch, specialised waves) against hfl_ln_qkv_fused +
the fp16 window kernel at the pyramid depths of the bench workload (Wild-Places cfg, 32 clouds) and of the CS-Wild-Places cfg
(K = 64): equality and timing.  `python tools/attn_ws_probe.py [cfg]`; HFL_WS_ABLATE=1 adds the timing ablations (needs a
library built with HFL_EXTRA_HIPCC_FLAGS=-DHFL_PROBES), HFL_WS_ONLY=1 runs the one kernel for tools/attn_ws_counters.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn  # noqa: E402
from hotformerloc_amd import _native  # noqa: E402
from hotformerloc_amd.plan import WindowPlan  # noqa: E402


def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main(cfg):
    params, depth = load_config(cfg)
    lib = _native.load()
    octree = build_batch_octree(syn.make_clouds(2, 32, 4096, params.coordinates), depth, 2, 'cuda')
    K = params.patch_size
    plan = WindowPlan(octree, K, 4, depth - 2, depth - 5, 3, 1, None)
    H, C = 16, 256
    g = torch.Generator(device='cuda').manual_seed(0)
    gamma = torch.rand(C, device='cuda', generator=g) + 0.5
    beta = torch.randn(C, device='cuda', generator=g) * 0.1
    w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.06
    b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
    qs = 16 ** -0.5 * 1.4426950408889634
    pack = ops.qkv_fused_pack(w)
    bnd = int(0.8 * K)
    table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.1
    for d in plan.pyramid_depths:
        nt, W = plan.n_tokens[d], plan.n_windows[d]
        x = torch.randn(nt + W, C, device='cuda', generator=g)
        qkv_all = torch.empty((nt + W, 3 * C), dtype=torch.float32, device='cuda')
        out = torch.zeros((nt + W, 2 * C), dtype=torch.bfloat16, device='cuda')

        def two(tab=table):
            ops.ln_qkv_fused(x, gamma, beta, 1e-5, pack, b, qs, out=qkv_all)
            return ops.window_attention(qkv_all, plan.meta[d], tab, nt, W, K, 1, 1, H, plan.B, rt_row0=nt, depth=d, out_split=2,
                                        qkv_f16=True)

        def relay_qkv():
            return ops.ln_qkv_fused(x[nt:], gamma, beta, 1e-5, pack, b, qs, out=qkv_all[nt:])

        def one():
            return ops.attn_ws(x[:nt], gamma, beta, 1e-5, pack, b, qs, qkv_all[nt:], plan.meta[d], table, nt, W, K, H,
                                     plan.B, d, out=out)
        if os.environ.get('HFL_WS_ONLY', '0') != '0':           # counter runs: the one kernel at the deepest level
            if d == plan.pyramid_depths[0]:
                if os.environ.get('HFL_WS_DBG'):
                    lib.hfl_set_variant(b'ws_dbg', int(os.environ['HFL_WS_DBG']))
                for _ in range(10):
                    one()
                torch.cuda.synchronize()
            continue
        # (the expanded table is cached per table tensor and consumer: the three-table form of the comparison gets its own copy)
        lib.hfl_set_variant(b'window_rpe_form1_max_depth', 0)
        a = two(table.clone()).clone()
        lib.hfl_set_variant(b'window_rpe_form1_max_depth', 4)
        f = one()
        nbad = (a[:nt].view(torch.int16) != f[:nt].view(torch.int16)).any(dim=1).sum().item()

        def val(t):
            v = t[nt:nt + W].float().view(W, C // 32, 2, 32)
            return (v[:, :, 0] + v[:, :, 1]).reshape(W, C)
        rerr = (val(a) - val(f)).abs().max().item()
        t2, t1, tr = timeit(two), timeit(one), timeit(relay_qkv)
        lib.hfl_set_variant(b'ws_map', 0)             # attention waves two per SIMD (map 1, the default: where the GEMM waves are not)
        f1 = one()
        same = torch.equal(f1.view(torch.int16), f.view(torch.int16))
        t1b = timeit(one)
        lib.hfl_set_variant(b'ws_map', 1)
        print('   wave map 0: %.1f us (map 1: %.1f), output identical: %s' % (t1b, t1, same), flush=True)
        if os.environ.get('HFL_WS_ABLATE', '0') != '0':        # timing ablations (wrong results): see WsParams::dbg
            abl = []
            for dbg in (1, 2, 3, 4, 7, 8):
                lib.hfl_set_variant(b'ws_dbg', dbg)
                abl.append((dbg, round(timeit(one), 1)))
            lib.hfl_set_variant(b'ws_dbg', 0)
            print('   ablations (bits: 1 no attention work, 2 no GEMM k-loop, 4 no weight stream, 8 no relay units) us:', abl, flush=True)
        flop = (6.0 * nt * C * C * 3 + 4.0 * (K + 1) * (K + 1) * C * W * 3.5)
        print('%s depth %d rows %d + %d relay: fused %.1f us (+ relay qkv %.1f us)  two launches %.1f us  x%.2f | token rows differ: %d, '
              'relay rows max err %.2e | %.0f TF/s issued, %.2f TB/s of x + out' % (cfg, d, nt, W, t1, tr, t2, t2 / (t1 + tr), nbad, rerr,
                                                                                   flop / t1 / 1e6, nt * C * 8 / t1 / 1e6), flush=True)


if __name__ == '__main__':
    for cfg in (sys.argv[1:] or ['wild-places']):
        main(cfg)
