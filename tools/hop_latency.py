"""Latency of a dependency between two HIP streams: event record + wait against a device flag (hfl_flag_set / hfl_flag_wait).
A chain of N hops: a short kernel on stream A, signal, stream B waits, a short kernel on B, signal back, ...; the time per hop
minus the kernel itself.  `python tools/hop_latency.py`"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hotformerloc_amd import _native  # noqa: E402

lib = _native.load()
dev = torch.device('cuda')
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 14, device=dev)
flag = torch.zeros(64, dtype=torch.int32, device=dev)
N = 200


def work(s):
    with torch.cuda.stream(s):
        x.add_(1.0)


def chain_events():
    for i in range(N):
        s, o = (sa, sb) if i % 2 == 0 else (sb, sa)
        work(s)
        ev = s.record_event()
        o.wait_event(ev)


def chain_flags(base):
    for i in range(N):
        s, o = (sa, sb) if i % 2 == 0 else (sb, sa)
        work(s)
        lib.hfl_flag_set(flag.data_ptr(), base + i + 1, ctypes.c_void_p(s.cuda_stream))
        lib.hfl_flag_wait(flag.data_ptr(), base + i + 1, 1 << 16, ctypes.c_void_p(o.cuda_stream))


def chain_one_stream():
    for i in range(N):
        work(sa)


big = torch.randn(8192, 8192, device=dev)


def timed(fn, *a):
    """GPU-side time of the chain: it is queued behind a ~30 ms blocker on both streams (so the host is done issuing before the
    first link runs) and bracketed by events on stream A (the chains end on A for even N)."""
    torch.cuda.synchronize()
    with torch.cuda.stream(sa):
        for _ in range(6):
            big @ big
        blk = sa.record_event()
    sb.wait_event(blk)
    with torch.cuda.stream(sa):
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    fn(*a)
    sa.wait_stream(sb)
    with torch.cuda.stream(sa):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3


base = 0
for rep in range(3):
    t1 = timed(chain_one_stream)
    te = timed(chain_events)
    tf = timed(chain_flags, base)
    base += N
    print('GPU time per link: same stream %.1f us | event record + wait %.1f us | flag set + wait kernels %.1f us' % (t1, te, tf), flush=True)
