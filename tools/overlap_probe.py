"""Would running two row-halves of a block on two streams beat one full-width chain?  MLP tail of a depth-4
H-OSA block (LN-split -> fc1 -> GELU-split -> fc2+residual) at M = 68k on one stream vs 2 x 34k on two."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hotformerloc_amd import ops

M, C = 68096, 256
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(M, C, device='cuda', generator=g)
w1 = ops.split_weight(torch.randn(4 * C, C, device='cuda', generator=g) * 0.05)
w2 = ops.split_weight(torch.randn(C, 4 * C, device='cuda', generator=g) * 0.05)
b1 = torch.zeros(4 * C, device='cuda'); b2 = torch.zeros(C, device='cuda')
gm = torch.ones(C, device='cuda'); bt = torch.zeros(C, device='cuda')

def chain(xs):
    h3 = ops.layer_norm_split3(xs, gm, bt, 1e-5)
    g3 = ops.bias_gelu_split3(ops.split_mm(h3, w1), b1)
    return ops.gemm_bf16(g3, w2, bias=b2, residual=xs)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
halves = [x[:M // 2], x[M // 2:]]

def one():
    return chain(x)

def two():
    main = torch.cuda.current_stream()
    outs = []
    for st, xs in zip((s1, s2), halves):
        st.wait_stream(main)
        with torch.cuda.stream(st):
            outs.append(chain(xs))
    main.wait_stream(s1); main.wait_stream(s2)
    return outs

def seq_halves():
    return [chain(xs) for xs in halves]

def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for name, fn in (('one stream, M=68k', one), ('two streams, 2 x 34k', two), ('one stream, 2 x 34k', seq_halves),
                 ('one stream, M=68k', one), ('two streams, 2 x 34k', two)):
    print('%-24s %.1f us' % (name, t(fn)))
