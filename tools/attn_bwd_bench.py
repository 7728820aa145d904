"""Window-attention backward at the bench shapes: first-generation kernel (fp32 MFMA, two softmax passes) vs the bf16 (hi, lo)
one-pass kernel; also checks that the two agree.  tools/attn_bwd_bench.py [cfg] [B] [points]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import build_batch_octree, load_config, ops, synthetic as syn, _native, autograd as ag
from hotformerloc_amd.plan import WindowPlan
cfg = sys.argv[1] if len(sys.argv) > 1 else 'wild-places'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
npts = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
params, depth = load_config(cfg)
octree = build_batch_octree(syn.make_clouds(2, B, npts, params.coordinates), depth, 2, 'cuda')
md = depth - 2
plan = WindowPlan(octree, params.patch_size, params.dilation, md, md - 3, 3, 1, params.ADaPE_mode)
K = params.patch_size
lib = _native.load()
lib.hfl_internal_set_window_bwd_rt.argtypes = [ctypes.c_int]
def timeit(fn, rounds=5, inner=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(rounds):
        e0.record()
        for _ in range(inner): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]
g = torch.Generator(device='cuda').manual_seed(0)
for d, H, G, dil in ((md, 8, 0, 1), (md, 8, 0, params.dilation), (md - 1, 16, 1, 1), (md - 2, 16, 1, 1), (md - 3, 16, 1, 1)):
    C = H * 16
    nt, W = plan.n_tokens[d], plan.n_windows[d]
    rows = nt + (W if G else 0)
    qkv = torch.randn(rows, 3 * C, device='cuda', generator=g)
    dout = torch.randn(rows, C, device='cuda', generator=g)
    bnd = int(0.8 * K * dil ** 0.5)
    table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.1
    desc = ag._desc(nt, W, K, dil, G, H, B, nt, d)
    res = {}
    for variant in (1, 2):
        lib.hfl_set_variant(b'window_bwd', variant)
        dqkv = torch.zeros_like(qkv); dtab = torch.zeros_like(table)
        def run():
            dtab.zero_()
            ops.check(lib.hfl_window_attention_bwd(dqkv.data_ptr(), dtab.data_ptr(), qkv.data_ptr(), dout.data_ptr(),
                      plan.meta[d].data_ptr(), table.data_ptr(), ctypes.byref(desc), ops._stream()), 'bwd')
        t = timeit(run)
        res[variant] = (t, dqkv.clone(), dtab.clone())
    # round 6: the table gradient on the matrix cores (default where the level's coordinates fit) against the LDS scatter-add
    lib.hfl_set_variant(b'window_bwd', 2)
    rt = {}
    for name, v in (('scatter-add', 0), ('matrix cores', -1)):
        lib.hfl_internal_set_window_bwd_rt(v)
        dqkv = torch.zeros_like(qkv); dtab = torch.zeros_like(table)
        def run2():
            dtab.zero_()
            ops.check(lib.hfl_window_attention_bwd(dqkv.data_ptr(), dtab.data_ptr(), qkv.data_ptr(), dout.data_ptr(),
                      plan.meta[d].data_ptr(), table.data_ptr(), ctypes.byref(desc), ops._stream()), 'bwd')
        rt[name] = (timeit(run2), dqkv.clone(), dtab.clone())
    lib.hfl_internal_set_window_bwd_rt(-1)
    print('   table gradient: scatter-add %.1f us, matrix cores %.1f us | dtable difference %.1e of its max, dqkv difference %.1e of its max'
          % (rt['scatter-add'][0], rt['matrix cores'][0],
             (rt['scatter-add'][2] - rt['matrix cores'][2]).abs().max().item() / rt['scatter-add'][2].abs().max().item(),
             (rt['scatter-add'][1] - rt['matrix cores'][1]).abs().max().item() / rt['scatter-add'][1].abs().max().item()))
    t_norpe = {}
    for variant in (1, 2):
        lib.hfl_set_variant(b'window_bwd', variant)
        dq2 = torch.zeros_like(qkv)
        t_norpe[variant] = timeit(lambda: ops.check(lib.hfl_window_attention_bwd(dq2.data_ptr(), None, qkv.data_ptr(), dout.data_ptr(),
                                  plan.meta[d].data_ptr(), None, ctypes.byref(desc), ops._stream()), 'bwd'))
    dq3 = torch.zeros_like(qkv)
    t_noflush = timeit(lambda: ops.check(lib.hfl_window_attention_bwd(dq3.data_ptr(), None, qkv.data_ptr(), dout.data_ptr(),
                       plan.meta[d].data_ptr(), table.data_ptr(), ctypes.byref(desc), ops._stream()), 'bwd'))
    print('   gen2 with the table but without the final global atomics: %.1f us' % t_noflush)
    lib.hfl_set_variant(b'window_bwd', 2)
    print('   without RPE table (no bias lookups, no table-gradient atomics): gen1 %.1f us, gen2 %.1f us' % (t_norpe[1], t_norpe[2]))
    e = (res[1][1] - res[2][1]).abs().max().item() / res[1][1].abs().max().item()
    et = (res[1][2] - res[2][2]).abs().max().item() / res[1][2].abs().max().item()
    tf = ops.window_attention(qkv, plan.meta[d], table, nt, W, K, dil, G, H, B, rt_row0=nt, depth=d)
    t_fwd = timeit(lambda: ops.window_attention(qkv, plan.meta[d], table, nt, W, K, dil, G, H, B, rt_row0=nt, depth=d))
    print('%s d=%d H=%d G=%d D=%d rows %d: bwd gen1 %.1f us  gen2 %.1f us (%.2fx)  forward (fp32 qkv) %.1f us | max diff dqkv %.1e dtable %.1e'
          % (cfg, d, H, G, dil, rows, res[1][0], res[2][0], res[1][0] / res[2][0], t_fwd, e, et))
