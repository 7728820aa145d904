"""Probe: hipBLASLt fp32 GEMM vs bf16 (3-term split, K-concatenated) GEMM with fp32 output."""
import torch, time
dev = 'cuda'
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(torch.__version__)
for (M, K, N) in [(68167, 256, 768), (68167, 256, 256), (68167, 256, 1024), (68167, 1024, 256), (118096, 128, 384), (118096, 128, 512), (118096, 512, 128), (14276, 256, 1024), (118096, 3456, 128)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    t32 = timeit(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * K * N
    xh = x.bfloat16(); xl = (x - xh.float()).bfloat16()
    wh = w.bfloat16(); wl = (w - wh.float()).bfloat16()
    A3 = torch.cat([xh, xh, xl], 1).contiguous(); W3 = torch.cat([wh, wl, wh], 1).contiguous()
    res = {}
    try:
        f = lambda: torch.mm(A3, W3.t(), out_dtype=torch.float32)
        y = f(); t3 = timeit(f)
        ref = torch.nn.functional.linear(x.double(), w.double())
        e3 = ((y.double() - ref).norm() / ref.norm()).item()
        e32 = ((torch.nn.functional.linear(x, w).double() - ref).norm() / ref.norm()).item()
        res['bf16x3 K-concat'] = (t3, e3)
        f1 = lambda: torch.mm(xh, wh.t(), out_dtype=torch.float32)
        res['bf16 single'] = (timeit(f1), 0)
    except Exception as ex:
        print('out_dtype path failed:', repr(ex)[:200]); e32 = 0
    try:
        fb = lambda: torch.mm(xh, wh.t())
        res['bf16->bf16'] = (timeit(fb), 0)
    except Exception as ex:
        print('bf16 mm failed', repr(ex)[:100])
    try:
        import sys, os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from hotformerloc_amd import ops
        if K % 32 == 0 and N % 128 == 0:
            x2, w2 = ops.split2(x), ops.split2_weight(w)
            fh = lambda: ops.linear_x3(x2, w2, bias=b)
            yh = fh(); th = timeit(fh)
            refb = torch.nn.functional.linear(x.double(), w.double(), b.double())
            res['hfl_linear_x3'] = (th, ((yh.double() - refb).norm() / refb.norm()).item())
    except Exception as ex:
        print('hfl linear failed', repr(ex)[:200])
    print('M=%d K=%d N=%d  fp32 %.3f ms (%.0f TF/s, err %.1e)' % (M, K, N, t32, fl / t32 / 1e9, e32), end='')
    for k, (t, e) in res.items():
        print(' | %s %.3f ms (%.0f TF/s eff%s)' % (k, t, fl / t / 1e9, (', err %.1e' % e) if e else ''), end='')
    print()
