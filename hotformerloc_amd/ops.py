"""Torch-facing wrappers over the C-ABI: tensors in, tensors out, current stream.

PyTorch is plumbing here (device memory + streams): every function hands raw
device pointers to libhotformerloc_hip.so on `torch.cuda.current_stream()` and
never synchronises.  Inputs must live on the GPU -- there is no CPU path.
"""

import ctypes
import weakref

import torch

from . import _native
from ._native import WindowAttnDesc, check


# torch.cuda.current_stream() / current_device() cost ~8 us / ~1 us of Python per call (device-index normalisation, lazy-init
# checks) and every launch wrapper needs both: 3.4 ms of a 9.9 ms host issue budget per forward.  The raw C hooks are the ones
# torch's own compiled-kernel launchers use.
_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_RAW_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _current_device() -> int:
    return _RAW_DEVICE() if _RAW_DEVICE is not None else torch.cuda.current_device()


def _stream():
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch HIP-event timing on the launch stream (bench.py roofline leg).
    `with KernelTimer() as t: ...; t.summary()` -> {kernel: (launches, ms, algorithmic bytes, flops, moved bytes)};
    "algorithmic" = SURVEY.md section 8(d)'s per-unit figure, "moved" = what this implementation actually
    reads + writes when that differs (e.g. the duplicated bf16 `hi` plane of a split output)."""
    active = None

    def __init__(self, only=None):
        self.records = []
        self.only = None if only is None else frozenset(only)     # time just these kernels (cheap: 2 events each)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, nbytes, flops, moved in self.records:
            n, ms, b, f, mv = out.get(name, (0, 0.0, 0, 0, 0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1), b + nbytes, f + flops, mv + moved)
        return out

    def by_size(self, name):
        """Launches of one kernel grouped by their algorithmic byte count: [(bytes per launch, launches, total ms)],
        largest first (a step's window-attention launches span 9 MB .. 280 MB: the small ones are latency-bound)."""
        torch.cuda.synchronize()
        groups = {}
        for n, e0, e1, nbytes, flops, moved in self.records:
            if n == name:
                c, ms = groups.get(nbytes, (0, 0.0))
                groups[nbytes] = (c + 1, ms + e0.elapsed_time(e1))
        return [(b, c, ms) for b, (c, ms) in sorted(groups.items(), reverse=True)]


class _timed:
    def __init__(self, name, nbytes, flops=0, moved=None):
        self.t = KernelTimer.active
        if self.t is not None and self.t.only is not None and name not in self.t.only:
            self.t = None
        if self.t is not None:
            self.name, self.nbytes, self.flops = name, int(nbytes), int(flops)
            self.moved = int(nbytes if moved is None else moved)

    def __enter__(self):
        if self.t is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            self.e1.record()
            self.t.records.append((self.name, self.e0, self.e1, self.nbytes, self.flops, self.moved))


def _dev(*tensors):
    """Every operand on the GPU, and on the CURRENT device: launches go to `torch.cuda.current_stream()` and the
    library's hipBLASLt state is per (device, stream), so a tensor of another GPU would be read through the wrong
    context.  Use `with torch.cuda.device(t.device):` around the model for a second GPU in one process."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _native.NativeLibraryError(
                'hotformerloc_amd ops need GPU tensors (got %s); there is no CPU fallback' % t.device)
        if cur is None:
            cur = _current_device()
        if t.device.index != cur:
            raise _native.NativeLibraryError(
                'tensor on %s but the current device is cuda:%d; wrap the call in torch.cuda.device(...)'
                % (t.device, cur))


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError('float32 expected, got %s' % t.dtype)
    return t.contiguous()


def _idx(neigh):
    if neigh.dtype == torch.int64:
        return neigh.contiguous(), 1
    if neigh.dtype == torch.int32:
        return neigh.contiguous(), 0
    raise TypeError('neighbour table must be int32 or int64, got %s' % neigh.dtype)


# ------------------------------------------------------------------ dwconv.core API
def dwconv_forward_backward(data: torch.Tensor, weight: torch.Tensor, neigh: torch.Tensor):
    """libs/dwconv/csrc/dwconv.h:13 -- data (N,C), weight (K,1,C), neigh (M,K) -> (M,C)."""
    _dev(data, weight, neigh)
    data, weight = _f32c(data), _f32c(weight)
    neigh, i64 = _idx(neigh)
    m, k = neigh.shape
    c = data.shape[1]
    out = torch.empty((m, c), dtype=torch.float32, device=data.device)
    check(_native.load().hfl_dwconv_forward_backward(
        out.data_ptr(), data.data_ptr(), weight.data_ptr(), neigh.data_ptr(), i64, m, c, k,
        _stream()), 'hfl_dwconv_forward_backward')
    return out


def dwconv_add(data: torch.Tensor, weight: torch.Tensor, neigh: torch.Tensor, add=None, out=None):
    """dwconv(data, weight, neigh) [+ add] through the CPE kernel's gather (hfl_dwconv_add): int32 table, C in {32, 64, 128,
    256}; the data gradient of CPE (neigh = the inverse table, add = the skip connection's gradient).  `out`: a preallocated
    contiguous (m, C) f32 view to write to (not `data`: the kernel gathers)."""
    _dev(data, weight, neigh, add, out)
    data, weight = _f32c(data), _f32c(weight)
    assert neigh.dtype == torch.int32 and neigh.is_contiguous()
    m, k = neigh.shape
    c = data.shape[1]
    if out is None:
        out = torch.empty((m, c), dtype=torch.float32, device=data.device)
    else:
        assert tuple(out.shape) == (m, c) and out.is_contiguous() and out.dtype == torch.float32
        assert out.data_ptr() != data.data_ptr()
    if add is not None:
        add = _f32c(add)
        assert tuple(add.shape) == (m, c)
    check(_native.load().hfl_dwconv_add(out.data_ptr(), data.data_ptr(), weight.data_ptr(), neigh.data_ptr(),
                                        None if add is None else add.data_ptr(), m, c, k, _stream()), 'hfl_dwconv_add')
    return out


def slot_sum(part: torch.Tensor, slot: torch.Tensor, bias=None):
    """out[h] = sum of part[slot[h, k]] over the row's live slots [+ bias] (hfl_slot_sum): the per-row sum of the partial
    products of a live-tap octree convolution.  part (P, C) f32, slot (n_out, K) int32 (-1 = none)."""
    _dev(part, slot, bias)
    part = _f32c(part)
    assert slot.dtype == torch.int32 and slot.is_contiguous() and slot.dim() == 2
    n, k = slot.shape
    c = part.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=part.device)
    check(_native.load().hfl_slot_sum(out.data_ptr(), part.data_ptr(), slot.data_ptr(),
                                      None if bias is None else _f32c(bias).data_ptr(), n, c, k, _stream()), 'hfl_slot_sum')
    return out


def dwconv_weight_backward(grad: torch.Tensor, data: torch.Tensor, neigh: torch.Tensor):
    """libs/dwconv/csrc/dwconv.h:14 -- -> (K,1,C)."""
    _dev(grad, data, neigh)
    grad, data = _f32c(grad), _f32c(data)
    neigh, i64 = _idx(neigh)
    n, k = neigh.shape
    c = data.shape[1]
    lib = _native.load()
    ws = torch.empty(int(lib.hfl_dwconv_weight_backward_workspace(n, c, k)), dtype=torch.uint8,
                     device=data.device)
    out = torch.empty((k, 1, c), dtype=torch.float32, device=data.device)
    check(lib.hfl_dwconv_weight_backward(out.data_ptr(), grad.data_ptr(), data.data_ptr(),
                                         neigh.data_ptr(), i64, n, c, k, ws.data_ptr(), _stream()),
          'hfl_dwconv_weight_backward')
    return out


def inverse_neigh(neigh: torch.Tensor):
    """libs/dwconv/csrc/dwconv.h:15."""
    _dev(neigh)
    neigh, i64 = _idx(neigh)
    out = torch.empty_like(neigh)
    check(_native.load().hfl_inverse_neigh(out.data_ptr(), neigh.data_ptr(), i64, neigh.shape[0],
                                           neigh.shape[1], _stream()), 'hfl_inverse_neigh')
    return out


def cpe_forward(x, weight, gamma, beta, neigh, residual: bool, eps: float = 1e-5, out=None, conv_out=None):
    """out = [x +] LayerNorm(dwconv(x, weight, neigh)) * gamma + beta (fused).  `out` may be a
    preallocated contiguous (n, C) view (e.g. the token rows of a larger buffer); `conv_out` (n, C) f32, optional:
    receives dwconv(x) (the training forward keeps it for the LayerNorm backward: hfl_cpe_forward_save)."""
    _dev(x, weight, gamma, beta, neigh, conv_out)
    x = _f32c(x)
    assert neigh.dtype == torch.int32 and neigh.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    else:
        assert out.shape == x.shape and out.is_contiguous() and out.dtype == torch.float32
        assert out.data_ptr() != x.data_ptr(), 'CPE gathers neighbours: cannot run in place'
    n, c = x.shape
    # algorithmic bytes: read x once, write out once, read the int32 neighbour rows
    with _timed('hfl_cpe_forward', n * c * 8 + n * neigh.shape[1] * 4, 2 * neigh.shape[1] * n * c):
        if conv_out is not None:
            assert conv_out.shape == x.shape and conv_out.is_contiguous() and conv_out.dtype == torch.float32
            check(_native.load().hfl_cpe_forward_save(
                out.data_ptr(), conv_out.data_ptr(), x.data_ptr(), _f32c(weight).data_ptr(), _f32c(gamma).data_ptr(),
                _f32c(beta).data_ptr(), neigh.data_ptr(), n, c, neigh.shape[1],
                float(eps), int(bool(residual)), _stream()), 'hfl_cpe_forward_save')
            return out
        check(_native.load().hfl_cpe_forward(
            out.data_ptr(), x.data_ptr(), _f32c(weight).data_ptr(), _f32c(gamma).data_ptr(),
            _f32c(beta).data_ptr(), neigh.data_ptr(), n, c, neigh.shape[1],
            float(eps), int(bool(residual)), _stream()), 'hfl_cpe_forward')
    return out


# ---------------------------------------------------------------------- layer norm
_LN_CHANNELS = (16, 32, 64, 128, 256, 512, 1024)


def layer_norm(x, weight, bias, eps: float = 1e-5):
    """LayerNorm over the last axis of a (..., C) fp32 tensor (HIP kernel, one pass)."""
    _dev(x, weight, bias)
    c = x.shape[-1]
    xc = _f32c(x)
    out = torch.empty_like(xc)                  # not a view: autograd Functions return it as it is
    with _timed('hfl_layer_norm', xc.numel() * 8):
        check(_native.load().hfl_layer_norm(out.data_ptr(), xc.data_ptr(), weight.data_ptr(),
                                            bias.data_ptr(), xc.numel() // c, c, float(eps), _stream()),
              'hfl_layer_norm')
    return out


def layer_norm_bwd(dy, x, weight, eps: float = 1e-5, dres=None):
    """(dx, dgamma, dbeta) of LayerNorm over the last axis (hfl_layer_norm_bwd; statistics recomputed from x); with
    `dres` the gradient of a skip connection around the normalised branch is added to dx in the same pass."""
    _dev(dy, x, weight, dres)
    c = x.shape[-1]
    x2, dy2 = _f32c(x).view(-1, c), _f32c(dy).view(-1, c)
    lib = _native.load()
    nb = int(lib.hfl_layer_norm_bwd_blocks(x2.shape[0], c))
    if nb <= 0:
        raise _native.NativeLibraryError('hfl_layer_norm_bwd: unsupported channel count %d' % c)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    part = torch.empty((2, nb, c), dtype=torch.float32, device=x.device)
    check(lib.hfl_layer_norm_bwd_add(dx.data_ptr(), part[0].data_ptr(), part[1].data_ptr(), dy2.data_ptr(), x2.data_ptr(),
                                     _f32c(weight).data_ptr(), None if dres is None else _f32c(dres).data_ptr(),
                                     x2.shape[0], c, float(eps), _stream()), 'hfl_layer_norm_bwd')
    dgb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    check(lib.hfl_layer_norm_bwd_finalize(dgb[0].data_ptr(), dgb[1].data_ptr(), part[0].data_ptr(), part[1].data_ptr(),
                                          nb, c, _stream()), 'hfl_layer_norm_bwd_finalize')
    return dx, dgb[0], dgb[1]


def add_layer_norm(x, y, weight, bias, eps: float = 1e-5, add_bias=None, inplace: bool = False):
    """(x + y [+ add_bias], LN(x + y [+ add_bias])) in one pass over the rows."""
    _dev(x, y, weight, bias, add_bias)
    c = x.shape[-1]
    x2, y2 = _f32c(x).view(-1, c), _f32c(y).view(-1, c)
    xo = x2 if inplace else torch.empty_like(x2)
    h = torch.empty_like(x2)
    with _timed('hfl_add_layer_norm', x2.numel() * 16):
        check(_native.load().hfl_add_layer_norm(
            xo.data_ptr(), h.data_ptr(), x2.data_ptr(), y2.data_ptr(),
            None if add_bias is None else add_bias.data_ptr(), weight.data_ptr(), bias.data_ptr(),
            x2.shape[0], c, float(eps), _stream()), 'hfl_add_layer_norm')
    return xo.view(x.shape), h.view(x.shape)


# ------------------------------------------------------- split-precision Linear path
def _a3(rows, c, device):
    return torch.empty((rows, 3 * c), dtype=torch.bfloat16, device=device)


def layer_norm_split3(x, weight, bias, eps: float = 1e-5):
    """A3 = [hi | hi | lo] bf16 of LN(x): the A operand of `split_mm` (header section 9)."""
    _dev(x, weight, bias)
    c = x.shape[-1]
    x2 = _f32c(x).view(-1, c)
    out = _a3(x2.shape[0], c, x.device)
    with _timed('hfl_layer_norm_split3', x2.numel() * 10):
        check(_native.load().hfl_layer_norm_split3(out.data_ptr(), x2.data_ptr(), weight.data_ptr(),
                                                   bias.data_ptr(), x2.shape[0], c, float(eps),
                                                   _stream()), 'hfl_layer_norm_split3')
    return out


def add_layer_norm_split3(x, y, weight, bias, eps: float = 1e-5, add_bias=None):
    """(x + y [+ add_bias] as fp32, split3(LN(that)))."""
    _dev(x, y, weight, bias, add_bias)
    c = x.shape[-1]
    x2, y2 = _f32c(x).view(-1, c), _f32c(y).view(-1, c)
    xo = torch.empty_like(x2)
    h = _a3(x2.shape[0], c, x.device)
    with _timed('hfl_add_layer_norm_split3', x2.numel() * 18):
        check(_native.load().hfl_add_layer_norm_split3(
            xo.data_ptr(), h.data_ptr(), x2.data_ptr(), y2.data_ptr(),
            None if add_bias is None else add_bias.data_ptr(), weight.data_ptr(), bias.data_ptr(),
            x2.shape[0], c, float(eps), _stream()), 'hfl_add_layer_norm_split3')
    return xo.view(x.shape), h


def add_bias(x, y, bias=None):
    """x + y + bias (fp32), one pass."""
    _dev(x, y, bias)
    c = x.shape[-1]
    x2, y2 = _f32c(x).view(-1, c), _f32c(y).view(-1, c)
    out = torch.empty_like(x2)
    with _timed('hfl_add_bias', x2.numel() * 12):
        check(_native.load().hfl_add_bias(out.data_ptr(), x2.data_ptr(), y2.data_ptr(),
                                          None if bias is None else bias.data_ptr(), x2.shape[0], c,
                                          _stream()), 'hfl_add_bias')
    return out.view(x.shape)


def bias_gelu_split3(x, bias=None):
    """split3(gelu(x + bias)) with the exact (erf) GELU."""
    _dev(x, bias)
    c = x.shape[-1]
    x2 = _f32c(x).view(-1, c)
    out = _a3(x2.shape[0], c, x.device)
    with _timed('hfl_bias_gelu_split3', x2.numel() * 10):
        check(_native.load().hfl_bias_gelu_split3(out.data_ptr(), x2.data_ptr(),
                                                  None if bias is None else bias.data_ptr(),
                                                  x2.shape[0], c, _stream()), 'hfl_bias_gelu_split3')
    return out


def split3(x):
    _dev(x)
    c = x.shape[-1]
    x2 = _f32c(x).view(-1, c)
    out = _a3(x2.shape[0], c, x.device)
    check(_native.load().hfl_split3(out.data_ptr(), x2.data_ptr(), x2.shape[0], c, _stream()),
          'hfl_split3')
    return out


def split_weight(w: torch.Tensor) -> torch.Tensor:
    """(N, K) fp32 Linear weight -> W3 = [w_hi | w_lo | w_hi] bf16 (N, 3K)."""
    w = w.detach().float()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def split_mm(a3: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    """fp32 (rows, N) = A3 @ W3^T: one hipBLASLt bf16 GEMM, fp32 accumulate and output,
    computing x_hi w_hi + x_hi w_lo + x_lo w_hi."""
    return torch.mm(a3, w3.t(), out_dtype=torch.float32)


def gemm_bf16(a3: torch.Tensor, w3: torch.Tensor, bias=None, residual=None, out=None) -> torch.Tensor:
    """fp32 (rows, N) = A3 @ W3^T [+ bias] [+ residual] in one hipBLASLt launch (hfl_gemm_bf16): the
    Linear + bias + residual add of a transformer block without a pass over the residual stream."""
    _dev(a3, w3, bias, residual)
    assert a3.dtype == torch.bfloat16 and w3.dtype == torch.bfloat16 and a3.is_contiguous() and w3.is_contiguous()
    m, k = a3.shape
    n = w3.shape[0]
    assert w3.shape[1] == k
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a3.device)
    else:
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (m, n)
    if residual is not None:
        residual = _f32c(residual)
        assert tuple(residual.shape) == (m, n)
    check(_native.load().hfl_gemm_bf16(out.data_ptr(), a3.data_ptr(), w3.data_ptr(),
                                       None if bias is None else _f32c(bias).data_ptr(),
                                       None if residual is None else residual.data_ptr(), m, n, k, _stream()),
          'hfl_gemm_bf16')
    return out


# ------------------------------------------- hand-written split-precision Linear (csrc/gemm_x3.hip)
def split2(x, row_scale=None):
    """fp32 (rows, C) [* row_scale (rows)] -> split2 bf16 (rows, 2C): per 32-channel block [32 x hi | 32 x lo]
    (hfl_split2 / hfl_split2_rows)."""
    _dev(x, row_scale)
    c = x.shape[-1]
    x2 = _f32c(x).view(-1, c)
    out = torch.empty((x2.shape[0], 2 * c), dtype=torch.bfloat16, device=x.device)
    check(_native.load().hfl_split2_rows(out.data_ptr(), x2.data_ptr(),
                                         None if row_scale is None else _f32c(row_scale).data_ptr(), x2.shape[0], c,
                                         _stream()), 'hfl_split2')
    return out


def layer_norm_split2(x, weight, bias, eps: float = 1e-5):
    """split2(LN(x)): the operand of `linear_x3` (hfl_layer_norm_split2)."""
    _dev(x, weight, bias)
    c = x.shape[-1]
    x2 = _f32c(x).view(-1, c)
    out = torch.empty((x2.shape[0], 2 * c), dtype=torch.bfloat16, device=x.device)
    with _timed('hfl_layer_norm_split2', x2.numel() * 8):
        check(_native.load().hfl_layer_norm_split2(out.data_ptr(), x2.data_ptr(), weight.data_ptr(),
                                                   bias.data_ptr(), x2.shape[0], c, float(eps),
                                                   _stream()), 'hfl_layer_norm_split2')
    return out


def layer_norm_relu(x, weight, bias, eps: float = 1e-5, split2: bool = False):
    """relu(LN(x)) in one pass (hfl_layer_norm_relu): fp32 rows, or -- `split2` -- the bf16 split2 operand (rows, 2C) of the
    next convolution's GEMM."""
    _dev(x, weight, bias)
    c = x.shape[-1]
    x2 = _f32c(x).view(-1, c)
    if split2:
        out = torch.empty((x2.shape[0], 2 * c), dtype=torch.bfloat16, device=x.device)
    else:
        out = torch.empty_like(x2)
    if x2.shape[0] == 0:
        return out
    with _timed('hfl_layer_norm_relu', x2.numel() * 8):
        check(_native.load().hfl_layer_norm_relu(None if split2 else out.data_ptr(), out.data_ptr() if split2 else None,
                                                 x2.data_ptr(), weight.data_ptr(), bias.data_ptr(), x2.shape[0], c,
                                                 float(eps), _stream()), 'hfl_layer_norm_relu')
    return out


def split2_weight(w: torch.Tensor) -> torch.Tensor:
    """(N, K) fp32 Linear weight -> split2 bf16 (N, 2K) (host-side layout, once per parameter)."""
    w = w.detach().float()
    n, k = w.shape
    assert k % 32 == 0
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.view(n, k // 32, 32), lo.view(n, k // 32, 32)], dim=2).reshape(n, 2 * k).contiguous()


def linear_x3(x2: torch.Tensor, w2: torch.Tensor, bias=None, residual=None, gelu_split_out: bool = False,
              out=None, row_scale=None) -> torch.Tensor:
    """y = x W^T [+ bias] [+ residual] (fp32), or with gelu_split_out the split2 bf16 operand of the next Linear,
    split2(gelu(x W^T + bias)); x2 / w2 are split2 operands (hfl_linear_x3)."""
    _dev(x2, w2, bias, *(residual if isinstance(residual, (list, tuple)) else (residual,)))
    assert x2.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and x2.is_contiguous() and w2.is_contiguous()
    m, k2 = x2.shape
    n = w2.shape[0]
    assert w2.shape[1] == k2 and k2 % 64 == 0
    k = k2 // 2
    if gelu_split_out:
        assert residual is None
        if out is None:
            out = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=x2.device)
    else:
        if out is None:
            out = torch.empty((m, n), dtype=torch.float32, device=x2.device)
        if isinstance(residual, (list, tuple)):       # the residual rows in several arrays (hfl_linear_x3_seg)
            assert row_scale is None
            seg, rows = row_segments(residual)
            assert rows == m and residual[0].shape[1] == n
            with _timed('hfl_linear_x3', m * k * 4 + m * n * 8, 2 * m * k * n):
                check(_native.load().hfl_linear_x3_seg(out.data_ptr(), x2.data_ptr(), w2.data_ptr(),
                                                       None if bias is None else _f32c(bias).data_ptr(), ctypes.byref(seg),
                                                       m, k, n, _stream()), 'hfl_linear_x3_seg')
            return out
        if residual is not None:
            residual = _f32c(residual)
            assert tuple(residual.shape) == (m, n)
    # algorithmic bytes: x once (4 B/elt), out once (4 B/elt either form), residual once; 2 M K N flop (fp32-equivalent)
    with _timed('hfl_linear_x3', m * k * 4 + m * n * (8 if residual is not None else 4), 2 * m * k * n):
        if row_scale is not None:          # (acc + bias) * row_scale[m] + residual: stochastic depth of a fused branch
            assert not gelu_split_out and tuple(row_scale.shape) == (m,)
            check(_native.load().hfl_linear_x3_rows(out.data_ptr(), x2.data_ptr(), w2.data_ptr(),
                                                    None if bias is None else _f32c(bias).data_ptr(),
                                                    None if residual is None else residual.data_ptr(),
                                                    _f32c(row_scale).data_ptr(), m, k, n, _stream()), 'hfl_linear_x3_rows')
            return out
        check(_native.load().hfl_linear_x3(out.data_ptr(), x2.data_ptr(), w2.data_ptr(),
                                           None if bias is None else _f32c(bias).data_ptr(),
                                           None if residual is None else residual.data_ptr(), m, k, n,
                                           int(bool(gelu_split_out)), _stream()), 'hfl_linear_x3')
    return out


def x6_pack(w: torch.Tensor) -> torch.Tensor:
    """(N, K) fp32 Linear weight -> its three bf16 planes (3, N, Kp) for `linear_x6` (hfl_linear_x6_pack, once per parameter;
    Kp = K rounded up to a multiple of 64, zero-padded)."""
    _dev(w)
    wc = _f32c(w.detach())
    n, k = wc.shape
    assert k % 32 == 0 and n % 128 == 0
    w3 = torch.empty((3, n, (k + 63) // 64 * 64), dtype=torch.bfloat16, device=w.device)
    check(_native.load().hfl_linear_x6_pack(w3.data_ptr(), wc.data_ptr(), n, k, _stream()), 'hfl_linear_x6_pack')
    return w3


def linear_x6_ok(in_features: int, out_features: int) -> bool:
    return in_features % 32 == 0 and out_features % 128 == 0 and 2 * (in_features + 32) * out_features < 2 ** 31


def linear_x6(x: torch.Tensor, w3: torch.Tensor, bias=None, residual=None, gelu: bool = False, out=None,
              row_scale=None) -> torch.Tensor:
    """y = x W^T [+ bias] [gelu] [* row_scale] [+ residual], fp32 in and out, fp32-grade products (hfl_linear_x6: three
    bf16 planes per operand, six plane products with fp32 accumulation).  x (M, K) f32, w3 = x6_pack(W)."""
    _dev(x, w3, bias, residual, row_scale)
    assert w3.dtype == torch.bfloat16 and w3.is_contiguous() and w3.dim() == 3 and w3.shape[0] == 3
    xc = _f32c(x)
    m, k = xc.shape
    n = w3.shape[1]
    assert w3.shape[2] == (k + 63) // 64 * 64
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    else:
        assert out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (m, n)
    if residual is not None:
        residual = _f32c(residual)
        assert tuple(residual.shape) == (m, n)
    if row_scale is not None:
        row_scale = _f32c(row_scale)
        assert tuple(row_scale.shape) == (m,)
    with _timed('hfl_linear_x6', m * k * 4 + m * n * (8 if residual is not None else 4), 2 * m * k * n):
        check(_native.load().hfl_linear_x6(out.data_ptr(), xc.data_ptr(), w3.data_ptr(),
                                           None if bias is None else _f32c(bias).data_ptr(),
                                           None if residual is None else residual.data_ptr(),
                                           None if row_scale is None else row_scale.data_ptr(), m, k, n, int(bool(gelu)),
                                           _stream()), 'hfl_linear_x6')
    return out


def linear_x6_grouped_gather(x: torch.Tensor, src: torch.Tensor, w3: torch.Tensor, tiles: torch.Tensor,
                             out_features: int) -> torch.Tensor:
    """The per-tap products of an octree convolution over its live (row, tap) pairs at matched precision: row m of the
    operand is x[src[m]] (x (n_src, Cin) f32), tiles (n_tiles, 3) int32 as `linear_x3_grouped`, w3 = x6_pack of the stacked
    per-tap weight blocks (each padded to a multiple of 128 rows); returns (len(src), out_features) f32."""
    _dev(x, src, w3, tiles)
    xc = _f32c(x)
    assert src.dtype == torch.int32 and src.is_contiguous() and tiles.dtype == torch.int32 and tiles.is_contiguous()
    assert w3.dtype == torch.bfloat16 and w3.is_contiguous() and w3.dim() == 3 and w3.shape[0] == 3
    k = xc.shape[1]
    assert w3.shape[2] == (k + 63) // 64 * 64
    m = src.shape[0]
    out = torch.empty((m, out_features), dtype=torch.float32, device=x.device)
    with _timed('hfl_linear_x6', m * k * 4 + m * out_features * 4, 2 * m * k * out_features):
        check(_native.load().hfl_linear_x6_grouped_gather(out.data_ptr(), xc.data_ptr(), src.data_ptr(), w3.data_ptr(),
                                                          w3.shape[1], tiles.data_ptr(), tiles.shape[0], m, k, out_features,
                                                          _stream()), 'hfl_linear_x6_grouped_gather')
    return out


def linear_x3_gelu_fwd(x2: torch.Tensor, w2: torch.Tensor, bias):
    """(split2(gelu(x W^T + b)), x W^T + b as f32) in one launch (hfl_linear_x3_gelu_fwd): training forward of fc1."""
    _dev(x2, w2, bias)
    assert x2.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and x2.is_contiguous() and w2.is_contiguous()
    m, k2 = x2.shape
    n = w2.shape[0]
    assert w2.shape[1] == k2
    out = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=x2.device)
    pre = torch.empty((m, n), dtype=torch.float32, device=x2.device)
    with _timed('hfl_linear_x3', m * (k2 // 2) * 4 + m * n * 8, 2 * m * (k2 // 2) * n):
        check(_native.load().hfl_linear_x3_gelu_fwd(out.data_ptr(), pre.data_ptr(), x2.data_ptr(), w2.data_ptr(),
                                                    None if bias is None else _f32c(bias).data_ptr(), m, k2 // 2, n,
                                                    _stream()), 'hfl_linear_x3_gelu_fwd')
    return out, pre


def linear_x3_gelu_bwd(dy2: torch.Tensor, wt2: torch.Tensor, preact: torch.Tensor):
    """split2((dy W) * gelu'(preact)) (hfl_linear_x3_gelu_bwd): wt2 = split2 of W^T, preact (rows, wt2.shape[0]) f32."""
    _dev(dy2, wt2, preact)
    assert dy2.dtype == torch.bfloat16 and wt2.dtype == torch.bfloat16 and dy2.is_contiguous() and wt2.is_contiguous()
    m, k2 = dy2.shape
    n = wt2.shape[0]
    assert wt2.shape[1] == k2 and tuple(preact.shape) == (m, n) and preact.is_contiguous()
    out = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=dy2.device)
    with _timed('hfl_linear_x3', m * (k2 // 2) * 4 + m * n * 8, 2 * m * (k2 // 2) * n):
        check(_native.load().hfl_linear_x3_gelu_bwd(out.data_ptr(), dy2.data_ptr(), wt2.data_ptr(), preact.data_ptr(),
                                                    m, k2 // 2, n, _stream()), 'hfl_linear_x3_gelu_bwd')
    return out


def linear_x3_grouped(x2: torch.Tensor, w2: torch.Tensor, tiles: torch.Tensor, out_features: int) -> torch.Tensor:
    """One launch of the split-precision GEMM over row tiles with different weight blocks (hfl_linear_x3_grouped):
    x2 (rows, 2K) split2, w2 (blocks * Npad, 2K) split2, tiles (n, 3) int32 {first row, rows, first weight row};
    returns (rows, out_features) f32."""
    _dev(x2, w2, tiles)
    assert x2.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and x2.is_contiguous() and w2.is_contiguous()
    assert tiles.dtype == torch.int32 and tiles.is_contiguous() and tiles.shape[1] == 3 and w2.shape[1] == x2.shape[1]
    m, k = x2.shape[0], x2.shape[1] // 2
    out = torch.empty((m, out_features), dtype=torch.float32, device=x2.device)
    with _timed('hfl_linear_x3', m * k * 4 + m * out_features * 4, 2 * m * k * out_features):
        check(_native.load().hfl_linear_x3_grouped(out.data_ptr(), x2.data_ptr(), w2.data_ptr(), tiles.data_ptr(),
                                                   tiles.shape[0], m, k, out_features, _stream()), 'hfl_linear_x3_grouped')
    return out


def linear_x3_grouped_gather(x2: torch.Tensor, src: torch.Tensor, w2: torch.Tensor, tiles: torch.Tensor,
                             out_features: int) -> torch.Tensor:
    """`linear_x3_grouped` over the rows x2[src[m]] without materialising them (hfl_linear_x3_grouped_gather): x2 (N, 2K) split2
    input rows of an octree convolution, src (P) or (P, 1) int32 the input row of every live pair; returns (P, out_features)."""
    _dev(x2, src, w2, tiles)
    assert x2.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and x2.is_contiguous() and w2.is_contiguous()
    assert tiles.dtype == torch.int32 and tiles.is_contiguous() and tiles.shape[1] == 3 and w2.shape[1] == x2.shape[1]
    assert src.dtype == torch.int32 and src.is_contiguous()
    m, k = src.numel(), x2.shape[1] // 2
    out = torch.empty((m, out_features), dtype=torch.float32, device=x2.device)
    with _timed('hfl_linear_x3', m * k * 4 + m * out_features * 4, 2 * m * k * out_features):
        check(_native.load().hfl_linear_x3_grouped_gather(out.data_ptr(), x2.data_ptr(), src.data_ptr(), x2.shape[0],
                                                          w2.data_ptr(), tiles.data_ptr(), tiles.shape[0], m, k, out_features,
                                                          _stream()), 'hfl_linear_x3_grouped_gather')
    return out


def mlp_fused_ok(channels: int) -> bool:
    """Channel widths `ln_mlp_fused` takes (hfl_mlp_fused_pack_bytes > 0)."""
    return int(_native.load().hfl_mlp_fused_pack_bytes(int(channels))) > 0


def mlp_fused_shape_ok(channels: int, hidden: int) -> bool:
    return int(_native.load().hfl_mlp_fused_pack_bytes_h(int(channels), int(hidden))) > 0


def mlp_fused_pack(w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """Weight image of `ln_mlp_fused` from the fp32 Linear weights fc1 (H, C) and fc2 (C, H) (hfl_mlp_fused_pack_h): the
    (hi, lo) bf16 split of both, cut into the 32-hidden-feature stages the kernel streams through LDS.  Once per parameter.
    H = 4C (C = 128, 256: transformer blocks) or H = C = 256 (Mixer layers)."""
    _dev(w1, w2)
    w1, w2 = _f32c(w1.detach()), _f32c(w2.detach())
    h, c = w1.shape
    assert tuple(w2.shape) == (c, h)
    lib = _native.load()
    n = int(lib.hfl_mlp_fused_pack_bytes_h(c, h))
    if n <= 0:
        raise _native.NativeLibraryError('hfl_mlp_fused_pack: unsupported shape C = %d, hidden = %d' % (c, h))
    pack = torch.empty(n, dtype=torch.uint8, device=w1.device)
    check(lib.hfl_mlp_fused_pack_h(pack.data_ptr(), w1.data_ptr(), w2.data_ptr(), c, h, _stream()), 'hfl_mlp_fused_pack_h')
    return pack


def ln_mlp_fused(x, gamma, beta, eps: float, pack, b1, b2, out=None):
    """out = x + fc2(gelu(fc1(LN(x)) + b1)) + b2 in ONE launch (hfl_ln_mlp_fused_h); `pack` from `mlp_fused_pack`; the hidden
    width is the length of b1."""
    _dev(x, gamma, beta, pack, b1, b2)
    x = _f32c(x)
    m, c = x.shape
    h = b1.numel()
    if out is None:
        out = torch.empty_like(x)
    else:
        assert out.shape == x.shape and out.dtype == torch.float32 and out.is_contiguous()
    assert out.data_ptr() != x.data_ptr(), 'the residual rows are re-read: cannot run in place'
    # algorithmic bytes: read x, write out (the LayerNorm input doubles as the residual); 2 * 2 * M * C * H flop
    lib = _native.load()
    assert pack.numel() == int(lib.hfl_mlp_fused_pack_bytes_h(c, h)), 'pack does not belong to this (C, hidden)'
    ws_bytes = int(lib.hfl_ln_mlp_fused_workspace_h(m, c, h))   # partial sums of the rows left over after the last whole round
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes > 0 else None
    with _timed('hfl_ln_mlp_fused', m * c * 8, 4 * m * c * h):
        check(lib.hfl_ln_mlp_fused_h(out.data_ptr(), x.data_ptr(), _f32c(gamma).data_ptr(), _f32c(beta).data_ptr(),
                                     float(eps), pack.data_ptr(), _f32c(b1).data_ptr(), _f32c(b2).data_ptr(), m, c, h,
                                     ws.data_ptr() if ws is not None else None, ws_bytes, _stream()), 'hfl_ln_mlp_fused_h')
    return out


def block_forward_x3(weights, keep_alive, x_in, relay, neigh, tok_meta, n_tokens: int, desc: WindowAttnDesc):
    """One transformer block of the inference path in ONE native call (hfl_block_forward_x3): CPE -> [relay rows] -> LN1 ->
    qkv -> window attention -> proj + residual -> LN2 -> fc1 + GELU -> fc2 + residual.  `weights` = a filled
    `_native.BlockWeights` (see model._block_weights), `keep_alive` the tensors its pointers refer to."""
    call = BlockCall(weights, keep_alive, x_in, neigh, tok_meta, n_tokens, desc)
    return call.run(0, relay)


class BlockCall:
    """The same block in two phases (hfl_block_io.phase): `run(1)` issues what reads token rows only (CPE, their LN1 and
    qkv projection) -- the caller may do that while the relay-token self-attention of the iteration is still running on
    another stream -- `run(2, relay)` the rest, on whatever stream is current at that call; `run(0, relay)` = both.
    `run(3, relay)` / `block_attention_multi([...])` / `run(4)` split phase 2 around the window attention, so that the
    attention of several blocks goes out as one launch."""

    def __init__(self, weights, keep_alive, x_in, neigh, tok_meta, n_tokens: int, desc: WindowAttnDesc):
        _dev(x_in, neigh, tok_meta)
        rows, c = x_in.shape
        self.lib = _native.load()
        self.weights, self.keep, self.desc = weights, (keep_alive, x_in, neigh, tok_meta), desc
        self.out = torch.empty((rows, c), dtype=torch.float32, device=x_in.device)
        self.arena = torch.empty(int(self.lib.hfl_block_forward_x3_arena(rows, c)), dtype=torch.uint8, device=x_in.device)
        self.io = _native.BlockIO(x_in=x_in.data_ptr(), relay=None, out=self.out.data_ptr(), arena=self.arena.data_ptr(),
                                  neigh=neigh.data_ptr(), tok_meta=tok_meta.data_ptr(), n_rows=rows, n_tokens=n_tokens,
                                  phase=0)

    def run(self, phase: int, relay=None):
        if relay is not None:           # (stays set for the later phases of the call: phase 4's proj reads the relay rows there)
            _dev(relay)
            assert relay.dtype == torch.float32 and relay.is_contiguous()
            self.keep = self.keep + (relay,)
            self.io.relay = relay.data_ptr()
        self.io.phase = phase
        check(self.lib.hfl_block_forward_x3(ctypes.byref(self.weights), ctypes.byref(self.io), ctypes.byref(self.desc),
                                            _stream()), 'hfl_block_forward_x3')
        return self.out


def block_attention_multi(calls):
    """The window attention of several blocks that have run phases 1 and 3 (`BlockCall.run(1)`, `.run(3, relay)`), as one
    launch when their attention shapes agree (hfl_block_attention_x3_multi); `.run(4)` of each block follows."""
    n = len(calls)
    assert 1 <= n <= 4
    lib = _native.load()
    ws = _native.ptr_array([ctypes.addressof(c.weights) for c in calls])
    ios = _native.ptr_array([ctypes.addressof(c.io) for c in calls])
    ds = _native.ptr_array([ctypes.addressof(c.desc) for c in calls])
    check(lib.hfl_block_attention_x3_multi(n, ws, ios, ds, _stream()), 'hfl_block_attention_x3_multi')


def row_segments(parts):
    """hfl_row_segments of the row-wise concatenation of `parts` (1..4 contiguous f32 (rows_i, C) tensors): (struct, rows)."""
    assert 1 <= len(parts) <= 4
    _dev(*parts)
    seg = _native.RowSegments()
    seg.n = len(parts)
    r = 0
    for i, t in enumerate(parts):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == parts[0].shape[1]
        seg.ptr[i], seg.row0[i] = t.data_ptr(), r
        r += t.shape[0]
    return seg, r


def relay_block_forward_x3(weights, keep_alive, rt, seq_rows, seq_off, batch: int, max_seq_len: int, orphan_rows=None):
    """The relay-token transformer block (RTSA) of the inference path in ONE native call (hfl_relay_block_forward_x3).  rt: the
    (rows, C) relay rows, or a list of up to four tensors whose row-wise concatenation they are (the pyramid levels' relay rows
    where the levels left them: needs weights.qkv_pack)."""
    seg = None
    if isinstance(rt, (list, tuple)):
        seg, rows = row_segments(rt)
        c, dev, x_ptr = rt[0].shape[1], rt[0].device, None
    else:
        _dev(rt)
        (rows, c), dev, x_ptr = rt.shape, rt.device, rt.data_ptr()
    _dev(seq_rows, seq_off, orphan_rows)
    lib = _native.load()
    out = torch.empty((rows, c), dtype=torch.float32, device=dev)
    arena = torch.empty(int(lib.hfl_relay_block_forward_x3_arena(rows, c)), dtype=torch.uint8, device=dev)
    io = _native.RelayBlockIO(x_in=x_ptr, x_segments=None if seg is None else ctypes.addressof(seg),
                              out=out.data_ptr(), arena=arena.data_ptr(), seq_rows=seq_rows.data_ptr(),
                              seq_off=seq_off.data_ptr(), n_rows=rows, batch=batch, max_seq_len=max_seq_len,
                              orphan_rows=None if orphan_rows is None or orphan_rows.numel() == 0 else orphan_rows.data_ptr(),
                              n_orphans=0 if orphan_rows is None else orphan_rows.numel())
    check(lib.hfl_relay_block_forward_x3(ctypes.byref(weights), ctypes.byref(io), _stream()), 'hfl_relay_block_forward_x3')
    return out


def wgrad_x3(dy2: torch.Tensor, x2: torch.Tensor, with_bias: bool = False):
    """(dW, db) of y = x W^T + b from split2 operands: dW (N, K) = dy^T x, db (N) = dy summed over rows (hfl_wgrad_x3;
    fixed reduction order)."""
    _dev(dy2, x2)
    assert dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and dy2.is_contiguous() and x2.is_contiguous()
    m = dy2.shape[0]
    n, k = dy2.shape[1] // 2, x2.shape[1] // 2
    assert x2.shape[0] == m and m > 0
    lib = _native.load()
    nbytes = int(lib.hfl_wgrad_x3_workspace(m, n, k))
    if nbytes <= 0:
        raise _native.NativeLibraryError('hfl_wgrad_x3: unsupported shape (%d, %d, %d)' % (m, n, k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy2.device)
    dw = torch.empty((n, k), dtype=torch.float32, device=dy2.device)
    db = torch.empty(n, dtype=torch.float32, device=dy2.device) if with_bias else None
    with _timed('hfl_wgrad_x3', m * (n + k) * 4, 2 * m * n * k):
        check(lib.hfl_wgrad_x3(dw.data_ptr(), None if db is None else db.data_ptr(), dy2.data_ptr(), x2.data_ptr(),
                               m, n, k, ws.data_ptr(), _stream()), 'hfl_wgrad_x3')
    return dw, db


def linear_x3_qkv(x2: torch.Tensor, w2: torch.Tensor, bias, q_scale: float, out=None) -> torch.Tensor:
    """qkv projection into the fp16 (hi, lo) operand layout of the v5 window-attention kernel (hfl_linear_x3_qkv):
    returns an opaque (rows, 3C) float32-sized buffer for `window_attention(..., qkv_f16=True)`."""
    _dev(x2, w2, bias)
    assert x2.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and x2.is_contiguous() and w2.is_contiguous()
    m, k2 = x2.shape
    n = w2.shape[0]
    assert w2.shape[1] == k2 and n % 3 == 0
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=x2.device)
    else:
        assert tuple(out.shape) == (m, n) and out.dtype == torch.float32 and out.is_contiguous()
    with _timed('hfl_linear_x3', m * (k2 // 2) * 4 + m * n * 4, 2 * m * (k2 // 2) * n):
        check(_native.load().hfl_linear_x3_qkv(out.data_ptr(), x2.data_ptr(), w2.data_ptr(),
                                               None if bias is None else _f32c(bias).data_ptr(), m, k2 // 2, n,
                                               float(q_scale), _stream()), 'hfl_linear_x3_qkv')
    return out


def qkv_fused_pack(w_qkv: torch.Tensor) -> torch.Tensor:
    """Weight image of `ln_qkv_fused` from the fp32 qkv weight (3C, C) (hfl_qkv_fused_pack).  Once per parameter."""
    _dev(w_qkv)
    w = _f32c(w_qkv.detach())
    c = w.shape[1]
    assert tuple(w.shape) == (3 * c, c)
    lib = _native.load()
    n = int(lib.hfl_qkv_fused_pack_bytes(c))
    if n <= 0:
        raise _native.NativeLibraryError('hfl_qkv_fused_pack: unsupported channel count %d' % c)
    pack = torch.empty(n, dtype=torch.uint8, device=w.device)
    check(lib.hfl_qkv_fused_pack(pack.data_ptr(), w.data_ptr(), c, _stream()), 'hfl_qkv_fused_pack')
    return pack


def ln_qkv_fused(x, gamma, beta, eps: float, pack, bias, q_scale: float, out=None):
    """LayerNorm + qkv projection into the fp16 (hi, lo) operand layout of the window kernel in ONE launch
    (hfl_ln_qkv_fused = layer_norm_split2 + linear_x3_qkv): an opaque (rows, 3C) float32-sized buffer."""
    _dev(gamma, beta, pack, bias)
    if isinstance(x, (list, tuple)):                  # the rows in several arrays (hfl_ln_qkv_fused_seg)
        seg, m = row_segments(x)
        c = x[0].shape[1]
        if out is None:
            out = torch.empty((m, 3 * c), dtype=torch.float32, device=x[0].device)
        with _timed('hfl_ln_qkv_fused', m * c * 16, 6 * m * c * c):
            check(_native.load().hfl_ln_qkv_fused_seg(out.data_ptr(), ctypes.byref(seg), _f32c(gamma).data_ptr(),
                                                      _f32c(beta).data_ptr(), float(eps), pack.data_ptr(), _f32c(bias).data_ptr(),
                                                      float(q_scale), m, c, _stream()), 'hfl_ln_qkv_fused_seg')
        return out
    _dev(x)
    x = _f32c(x)
    m, c = x.shape
    if out is None:
        out = torch.empty((m, 3 * c), dtype=torch.float32, device=x.device)
    # algorithmic bytes: read x, write the operand rows; 2 M C 3C flop
    with _timed('hfl_ln_qkv_fused', m * c * 16, 6 * m * c * c):
        check(_native.load().hfl_ln_qkv_fused(out.data_ptr(), x.data_ptr(), _f32c(gamma).data_ptr(), _f32c(beta).data_ptr(),
                                              float(eps), pack.data_ptr(), _f32c(bias).data_ptr(), float(q_scale), m, c,
                                              _stream()), 'hfl_ln_qkv_fused')
    return out


def attn_fused_ok(n_tokens: int, n_windows: int, patch_size: int, dilation: int, n_relay: int, n_heads: int, depth: int,
                  channels: int, has_rpe: bool) -> bool:
    """Whether `attn_fused` (LayerNorm -> qkv -> window attention in one launch) takes this configuration."""
    desc = WindowAttnDesc(n_tokens=n_tokens, rt_row0=n_tokens, n_windows=n_windows, patch_size=patch_size, dilation=dilation,
                          n_relay=n_relay, n_heads=n_heads, pos_bnd=int(0.8 * patch_size * dilation ** 0.5),
                          batch_size=1, scale=16 ** -0.5, depth=depth)
    return bool(_native.load().hfl_attn_fused_ok(ctypes.byref(desc), int(channels), 1 if has_rpe else 0))


def attn_fused(x, gamma, beta, eps: float, qkv_pack, qkv_bias, q_scale: float, tok_meta, rpe_table, n_tokens: int,
               n_windows: int, patch_size: int, dilation: int, n_heads: int, batch_size: int, depth: int, out=None):
    """split2(window_attention(qkv(LayerNorm(x)))) in ONE launch (hfl_attn_fused_fwd): x (n_tokens, C) f32 -> (n_tokens, 2C)
    bf16, the operand of the proj GEMM -- bitwise what `ln_qkv_fused` + `window_attention(..., out_split=2, qkv_f16=True)` give."""
    _dev(x, gamma, beta, qkv_pack, qkv_bias, tok_meta, rpe_table)
    x = _f32c(x)
    c = x.shape[1]
    assert x.shape[0] >= n_tokens
    if out is None:
        out = torch.empty((x.shape[0], 2 * c), dtype=torch.bfloat16, device=x.device)
    desc = WindowAttnDesc(n_tokens=n_tokens, rt_row0=n_tokens, n_windows=n_windows, patch_size=patch_size, dilation=dilation,
                          n_relay=0, n_heads=n_heads, pos_bnd=int(0.8 * patch_size * dilation ** 0.5),
                          batch_size=batch_size, scale=16 ** -0.5, depth=depth)
    table_ptr, expanded = None, None
    if rpe_table is not None:
        expanded = rpe_expand(rpe_table, n_heads, desc.pos_bnd, depth, True)
        assert expanded is not None
        desc.rpe_expanded = expanded.data_ptr()
        rpe_table = _f32c(rpe_table)
        table_ptr = rpe_table.data_ptr()
    # algorithmic bytes: read x, write the split2 output (8 B per (row, channel)) + 8 B of metadata per token
    real_windows = -(-n_tokens // patch_size)
    with _timed('hfl_attn_fused_fwd', n_tokens * c * 8 + n_tokens * 8,
                6 * n_tokens * c * c + 4 * patch_size * patch_size * c * real_windows):
        check(_native.load().hfl_attn_fused_fwd(out.data_ptr(), x.data_ptr(), _f32c(gamma).data_ptr(), _f32c(beta).data_ptr(),
                                                float(eps), qkv_pack.data_ptr(), _f32c(qkv_bias).data_ptr(), float(q_scale),
                                                tok_meta.data_ptr(), table_ptr, ctypes.byref(desc), _stream()),
              'hfl_attn_fused_fwd')
    return out


def attn_ws_ok(n_tokens: int, n_windows: int, patch_size: int, n_heads: int, depth: int, channels: int) -> bool:
    """Whether `attn_ws` takes this relay-token block configuration."""
    desc = WindowAttnDesc(n_tokens=n_tokens, rt_row0=n_tokens, n_windows=n_windows, patch_size=patch_size, dilation=1,
                          n_relay=1, n_heads=n_heads, pos_bnd=int(0.8 * patch_size), batch_size=1, scale=16 ** -0.5, depth=depth)
    return bool(_native.load().hfl_attn_ws_ok(ctypes.byref(desc), int(channels)))


def attn_ws(x, gamma, beta, eps: float, qkv_pack, qkv_bias, q_scale: float, relay_qkv, tok_meta, rpe_table, n_tokens: int,
            n_windows: int, patch_size: int, n_heads: int, batch_size: int, depth: int, out=None):
    """split2(window_attention(qkv(LayerNorm(x)), relay q / k / v)) of a relay-token block in ONE launch (hfl_attn_ws_fwd):
    x (n_tokens, C) f32 token rows, relay_qkv (n_windows, 3C) the relay rows' fp16 (hi, lo) operand rows (`ln_qkv_fused` over
    them) -> (n_tokens + n_windows, 2C) bf16, the operand of the proj GEMM (relay rows at n_tokens + window)."""
    _dev(x, gamma, beta, qkv_pack, qkv_bias, relay_qkv, tok_meta, rpe_table)
    x = _f32c(x)
    c = x.shape[1]
    assert x.shape[0] >= n_tokens and relay_qkv.is_contiguous() and relay_qkv.shape[0] >= n_windows
    if out is None:
        out = torch.zeros((n_tokens + n_windows, 2 * c), dtype=torch.bfloat16, device=x.device)
    bnd = int(0.8 * patch_size)
    desc = WindowAttnDesc(n_tokens=n_tokens, rt_row0=n_tokens, n_windows=n_windows, patch_size=patch_size, dilation=1,
                          n_relay=1, n_heads=n_heads, pos_bnd=bnd, batch_size=batch_size, scale=16 ** -0.5, depth=depth)
    tables = None
    if rpe_table is not None:
        tables = rpe_expand(rpe_table, n_heads, bnd, depth, 2)
        assert tables is not None
    with _timed('hfl_attn_ws_fwd', n_tokens * c * 8 + n_tokens * 8 + n_windows * c * 16,
                6 * n_tokens * c * c + 4 * (patch_size + 1) ** 2 * c * n_windows):
        check(_native.load().hfl_attn_ws_fwd(out.data_ptr(), x.data_ptr(), _f32c(gamma).data_ptr(), _f32c(beta).data_ptr(),
                                             float(eps), qkv_pack.data_ptr(), _f32c(qkv_bias).data_ptr(), float(q_scale),
                                             relay_qkv.data_ptr(), tok_meta.data_ptr(),
                                             None if tables is None else tables.data_ptr(), ctypes.byref(desc), _stream()),
              'hfl_attn_ws_fwd')
    return out


def window_attention_f16_ok(n_rows: int, patch_size: int, dilation: int, n_relay: int, n_heads: int, depth: int) -> bool:
    """Whether the window kernel takes the fp16 (hi, lo) qkv layout for this launch (else: fp32 qkv)."""
    desc = WindowAttnDesc(n_tokens=n_rows, rt_row0=0, n_windows=1, patch_size=patch_size, dilation=dilation,
                          n_relay=n_relay, n_heads=n_heads, pos_bnd=int(0.8 * patch_size * dilation ** 0.5),
                          batch_size=1, scale=16 ** -0.5, depth=depth)
    return bool(_native.load().hfl_window_attention_f16_ok(ctypes.byref(desc), int(n_rows)))


def stack3(x: torch.Tensor, order: str) -> torch.Tensor:
    """(M, C) fp32 -> (3M, C) bf16 planes stacked along the rows: 'hhl' = [hi; hi; lo], 'hlh' = [hi; lo; hi]
    (operands of `gemm_bf16_tn`, the contraction runs over the rows)."""
    x = _f32c(x)
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, hi, lo] if order == 'hhl' else [hi, lo, hi], 0)


def gemm_bf16_tn(a_stack: torch.Tensor, b_stack: torch.Tensor) -> torch.Tensor:
    """fp32 (N, K) = a_stack^T @ b_stack over the stacked rows (hfl_gemm_bf16_tn): dW = dy^T x."""
    _dev(a_stack, b_stack)
    assert a_stack.dtype == torch.bfloat16 and b_stack.dtype == torch.bfloat16
    assert a_stack.is_contiguous() and b_stack.is_contiguous() and a_stack.shape[0] == b_stack.shape[0]
    r, n = a_stack.shape
    k = b_stack.shape[1]
    out = torch.empty((n, k), dtype=torch.float32, device=a_stack.device)
    check(_native.load().hfl_gemm_bf16_tn(out.data_ptr(), a_stack.data_ptr(), b_stack.data_ptr(), r, n, k,
                                          _stream()), 'hfl_gemm_bf16_tn')
    return out


# ------------------------------------------------------------------------- gather
def octree_gather(data, neigh):
    """(N,C),(M,K) int32 -> (M, K*C): ocnn octree2col with zero fill."""
    _dev(data, neigh)
    data = _f32c(data)
    assert neigh.dtype == torch.int32 and neigh.is_contiguous()
    m, k = neigh.shape
    c = data.shape[1]
    out = torch.empty((m, k * c), dtype=torch.float32, device=data.device)
    with _timed('hfl_octree_gather', data.numel() * 4 + m * k * c * 4 + m * k * 4):
        check(_native.load().hfl_octree_gather(out.data_ptr(), data.data_ptr(), neigh.data_ptr(), m, k,
                                               c, _stream()), 'hfl_octree_gather')
    return out


def inverse_table(inverse: torch.Tensor, table: torch.Tensor):
    """inverse (n_src, K) int32 of a gather table (n_dst, K) int32: inverse[table[m, k], k] = m, -1 elsewhere
    (hfl_inverse_table)."""
    _dev(inverse, table)
    assert table.dtype == torch.int32 and inverse.dtype == torch.int32 and table.is_contiguous() and inverse.is_contiguous()
    check(_native.load().hfl_inverse_table(inverse.data_ptr(), inverse.shape[0], table.data_ptr(), table.shape[0],
                                           table.shape[1], _stream()), 'hfl_inverse_table')
    return inverse


def tap_wgrad(g: torch.Tensor, dpart: torch.Tensor, chunks: torch.Tensor, tap_chunk_off: torch.Tensor, taps: int,
              g_rows=None, d_rows=None):
    """dW (taps, Cin, Cout) of a live-tap octree convolution: dW[k] = g_k^T dpart_k over the pairs of tap k
    (hfl_tap_wgrad; `chunks` / `tap_chunk_off` from Octree.sparse_taps_bwd).  g_rows / d_rows (P,) int32: pair p reads row
    g_rows[p] of g / d_rows[p] of dpart (hfl_tap_wgrad_gather) instead of row p of a pair-major copy."""
    _dev(g, dpart, chunks, tap_chunk_off, g_rows, d_rows)
    g, dpart = _f32c(g), _f32c(dpart)
    cin, cout = g.shape[1], dpart.shape[1]
    n_chunks = chunks.shape[0]
    for t in (g_rows, d_rows):
        assert t is None or (t.dtype == torch.int32 and t.is_contiguous())
    pairs = (g_rows if g_rows is not None else g).shape[0]
    dw = torch.empty((taps, cin, cout), dtype=torch.float32, device=g.device)
    ws = torch.empty((max(n_chunks, 1), cin, cout), dtype=torch.float32, device=g.device)
    with _timed('hfl_tap_wgrad', pairs * (cin + cout) * 4, 2 * pairs * cin * cout):
        check(_native.load().hfl_tap_wgrad_gather(dw.data_ptr(), g.data_ptr(), None if g_rows is None else g_rows.data_ptr(),
                                                  dpart.data_ptr(), None if d_rows is None else d_rows.data_ptr(),
                                                  chunks.data_ptr(), n_chunks, tap_chunk_off.data_ptr(), taps, cin, cout,
                                                  ws.data_ptr(), _stream()), 'hfl_tap_wgrad_gather')
    return dw


def tap_lists(table: torch.Tensor, edges_out: torch.Tensor = None):
    """Live-tap lists of a (rows, taps) int32 table (hfl_tap_lists): (src capacity rows*taps, slot (rows, taps),
    edges (taps+1) int32 on the device).  No host synchronisation; the caller narrows `src` once it knows
    edges[-1]."""
    _dev(table)
    assert table.dtype == torch.int32 and table.is_contiguous() and table.dim() == 2
    rows, taps = table.shape
    lib = _native.load()
    dev = table.device
    src = torch.empty(max(rows * taps, 1), dtype=torch.int32, device=dev)
    slot = torch.empty((rows, taps), dtype=torch.int32, device=dev)
    edges = torch.empty(taps + 1, dtype=torch.int32, device=dev) if edges_out is None else edges_out
    assert edges.dtype == torch.int32 and edges.numel() == taps + 1 and edges.is_contiguous()
    ws = torch.empty(int(lib.hfl_tap_lists_workspace(rows, taps)), dtype=torch.uint8, device=dev)
    check(lib.hfl_tap_lists(src.data_ptr(), slot.data_ptr(), edges.data_ptr(), table.data_ptr(), rows, taps,
                            ws.data_ptr(), _stream()), 'hfl_tap_lists')
    return src, slot, edges


def tap_lists_multi(tables, edges_outs):
    """`tap_lists` for up to 16 tables in three launches in all (hfl_tap_lists_multi): [(src, slot, edges)] per table."""
    import ctypes
    lib = _native.load()
    n = len(tables)
    assert 1 <= n <= 16 and len(edges_outs) == n
    _dev(*tables)
    dev = tables[0].device
    out, ws = [], []
    for t, e in zip(tables, edges_outs):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.dim() == 2
        rows, taps = t.shape
        assert e.dtype == torch.int32 and e.numel() == taps + 1 and e.is_contiguous()
        out.append((torch.empty(max(rows * taps, 1), dtype=torch.int32, device=dev),
                    torch.empty((rows, taps), dtype=torch.int32, device=dev), e))
        ws.append(torch.empty(int(lib.hfl_tap_lists_workspace(rows, taps)), dtype=torch.uint8, device=dev))
    ptr = lambda xs: (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])            # noqa: E731
    rows = (ctypes.c_int64 * n)(*[t.shape[0] for t in tables])
    taps = (ctypes.c_int32 * n)(*[t.shape[1] for t in tables])
    check(lib.hfl_tap_lists_multi(n, ptr([o[0] for o in out]), ptr([o[1] for o in out]), ptr([o[2] for o in out]),
                                  ptr(tables), rows, taps, ptr(ws), _stream()), 'hfl_tap_lists_multi')
    return out


def tap_tiles(edges_dev: torch.Tensor, n_tiles: int, taps: int, w_rows: int, tile_rows: int = 128):
    """(n_tiles, 3) int32 row-tile table of `linear_x3_grouped` from the device-side tap edges (hfl_tap_tiles)."""
    _dev(edges_dev)
    assert edges_dev.dtype == torch.int32 and edges_dev.numel() == taps + 1
    tiles = torch.empty((n_tiles, 3), dtype=torch.int32, device=edges_dev.device)
    if n_tiles > 0:
        check(_native.load().hfl_tap_tiles(tiles.data_ptr(), edges_dev.data_ptr(), taps, tile_rows, w_rows, _stream()),
              'hfl_tap_tiles')
    return tiles


def pad_index(row_off: torch.Tensor, batch: int, nmax: int):
    """(batch * nmax) int64 gather index of a ragged per-cloud row stream, sentinel = row_off[batch] (hfl_pad_index)."""
    _dev(row_off)
    assert row_off.dtype == torch.int64 and row_off.numel() == batch + 1
    out = torch.empty(batch * nmax, dtype=torch.int64, device=row_off.device)
    check(_native.load().hfl_pad_index(out.data_ptr(), row_off.data_ptr(), batch, nmax, _stream()), 'hfl_pad_index')
    return out


def pad_rows(x: torch.Tensor, row_off: torch.Tensor, batch: int, nmax: int):
    """(batch, nmax, C) zero-padded per-cloud copy of the ragged rows x (N, C) (hfl_pad_rows)."""
    _dev(x, row_off)
    x = _f32c(x)
    assert row_off.dtype == torch.int64 and row_off.numel() == batch + 1 and x.shape[1] % 4 == 0
    out = torch.empty((batch, nmax, x.shape[1]), dtype=torch.float32, device=x.device)
    check(_native.load().hfl_pad_rows(out.data_ptr(), x.data_ptr(), row_off.data_ptr(), batch, nmax, x.shape[1], _stream()),
          'hfl_pad_rows')
    return out


def mixer_tail(x: torch.Tensor, channel_w, channel_b, row_w, row_b):
    """(B, k_out * out_d) = flatten(row_proj(channel_proj(x^T)^T)) of the Mixer aggregator in one launch (hfl_mixer_tail);
    x (B, K, C), channel_w (k_out, K), row_w (out_d, C)."""
    _dev(x, channel_w, channel_b, row_w, row_b)
    x = _f32c(x)
    b, k, c = x.shape
    ko, d = channel_w.shape[0], row_w.shape[0]
    assert tuple(channel_w.shape) == (ko, k) and tuple(row_w.shape) == (d, c)
    out = torch.empty((b, ko * d), dtype=torch.float32, device=x.device)
    check(_native.load().hfl_mixer_tail(out.data_ptr(), x.data_ptr(), _f32c(channel_w).data_ptr(), _f32c(channel_b).data_ptr(),
                                        _f32c(row_w).data_ptr(), _f32c(row_b).data_ptr(), b, k, c, ko, d, _stream()),
          'hfl_mixer_tail')
    return out


def attn_pool_ok(channels: int) -> bool:
    return bool(_native.load().hfl_attn_pool_ok(int(channels)))


def attn_pool(x: torch.Tensor, row_off: torch.Tensor, query: torch.Tensor, batch: int, scale: float, out=None):
    """(batch, k, C) = per cloud softmax(scale * query x^T) x over the cloud's rows of the ragged x (N, C)
    (hfl_attn_pool: salsa.py:25-55).  `out`: a (batch, k, C) view, possibly a slice of a wider token matrix along dim 1."""
    _dev(x, row_off, query)
    x, query = _f32c(x), _f32c(query)
    assert row_off.dtype == torch.int64 and row_off.numel() == batch + 1 and query.shape[1] == x.shape[1]
    k, c = query.shape
    if out is None:
        out = torch.empty((batch, k, c), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.shape == (batch, k, c) and out.stride(2) == 1 and out.stride(1) == c
    lib = _native.load()
    nb = lib.hfl_attn_pool_workspace(batch, k, c, x.shape[0])
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=x.device)
    with _timed('hfl_attn_pool', x.numel() * 4 + out.numel() * 4, 4 * x.shape[0] * k * c):
        check(lib.hfl_attn_pool(out.data_ptr(), out.stride(0), x.data_ptr(), row_off.data_ptr(), query.data_ptr(), batch, k, c,
                                x.shape[0], float(scale), ws.data_ptr(), int(nb), _stream()), 'hfl_attn_pool')
    return out


# ---------------------------------------------------------------------- attention
_RPE2_CACHE = {}        # (id(table), depth) -> (weakref to table, version, expanded table)


def rpe_expand(rpe_table, n_heads: int, pos_bnd: int, depth: int, f16: bool = False):
    """Expanded relative-position table of the window kernels (hfl_window_rpe_expand) for the fp32-qkv kernel or, with
    `f16`, for the fp16 (hi, lo) one (the two take different forms at depth 5+), cached per (table, depth, consumer) and
    rebuilt when the table is modified in place.  None when the consumer has no expanded form for this depth."""
    lib = _native.load()
    n = lib.hfl_window_rpe_expand_size(n_heads, pos_bnd, depth, int(f16))
    if n <= 0:
        return None
    key = (id(rpe_table), depth, int(f16))            # (f16 = 2: the fused attention kernels' three 1-D tables at every depth)
    hit = _RPE2_CACHE.get(key)
    # data_ptr / device: `module.to(...)` swaps `.data` without bumping the version counter
    if (hit is not None and hit[0]() is rpe_table and hit[1] == rpe_table._version and hit[3] == rpe_table.data_ptr()
            and hit[2].device == rpe_table.device):
        return hit[2]
    out = torch.empty(n, dtype=torch.float32, device=rpe_table.device)
    src = _f32c(rpe_table.detach())
    check(lib.hfl_window_rpe_expand(out.data_ptr(), src.data_ptr(), n_heads, pos_bnd, depth, int(f16), _stream()),
          'hfl_window_rpe_expand')
    if len(_RPE2_CACHE) > 512:
        _RPE2_CACHE.clear()
    _RPE2_CACHE[key] = (weakref.ref(rpe_table), rpe_table._version, out, rpe_table.data_ptr())
    return out


def window_attention(qkv, tok_meta, rpe_table, n_tokens: int, n_windows: int, patch_size: int,
                     dilation: int, n_relay: int, n_heads: int, batch_size: int, rt_row0: int = 0,
                     depth: int = 0, qkv_bias=None, out_split: bool = False, qkv_f16: bool = False):
    """qkv (rows, 3*H*16) -> out (rows, H*16) fp32, or with out_split the bf16 [hi|hi|lo]
    (rows, 3*H*16) operand of the projection GEMM; qkv_bias is added to q,k,v on load.  qkv_f16: `qkv` is the
    fp16 (hi, lo) operand buffer of `linear_x3_qkv` (bias and softmax scale already folded in).
    See hfl_window_attention_fwd(_ex)."""
    _dev(qkv, tok_meta, rpe_table, qkv_bias)
    qkv = _f32c(qkv)
    rows = qkv.shape[0]
    c = n_heads * 16
    assert qkv.shape[1] == 3 * c
    out_split = int(out_split)          # 0: fp32 out; 1: bf16 [hi|hi|lo] (rows, 3c); 2: bf16 split2 (rows, 2c)
    if out_split:
        width = 3 * c if out_split == 1 else 2 * c
        # rows the kernel does not own (none today) must not hold NaN bit patterns
        out = torch.zeros((rows, width), dtype=torch.bfloat16, device=qkv.device) if rows > n_tokens + \
            (n_windows if n_relay else 0) else torch.empty((rows, width), dtype=torch.bfloat16, device=qkv.device)
    else:
        out = torch.empty((rows, c), dtype=torch.float32, device=qkv.device)
    desc = WindowAttnDesc(n_tokens=n_tokens, rt_row0=rt_row0, n_windows=n_windows,
                          patch_size=patch_size, dilation=dilation, n_relay=n_relay,
                          n_heads=n_heads, pos_bnd=int(0.8 * patch_size * dilation ** 0.5),
                          batch_size=batch_size, scale=16 ** -0.5, depth=depth)
    table_ptr = None
    if rpe_table is not None:
        assert tuple(rpe_table.shape) == (3 * (2 * desc.pos_bnd + 1), n_heads)
        expanded = rpe_expand(rpe_table, n_heads, desc.pos_bnd, depth, qkv_f16)   # keeps the tensor alive below
        if expanded is not None:
            desc.rpe_expanded = expanded.data_ptr()
        rpe_table = _f32c(rpe_table)
        table_ptr = rpe_table.data_ptr()
    used = n_tokens + (n_windows if n_relay else 0)          # rows the kernel touches
    seq = patch_size + n_relay
    real_windows = -(-n_tokens // patch_size)
    # algorithmic work: read q,k,v + write out = 16 B per (row, channel); QK^T + PV = 4 L^2 C per
    # window.  Moved: the bf16 [hi|hi|lo] output is 6 B instead of 4 B per channel, plus 8 B of metadata per token
    with _timed('hfl_window_attention_fwd', used * c * 16, 4 * seq * seq * c * real_windows,
                moved=used * c * (18 if out_split == 1 else 16) + n_tokens * 8):
        check(_native.load().hfl_window_attention_fwd_ex(
            out.data_ptr(), qkv.data_ptr(), None if qkv_bias is None else _f32c(qkv_bias).data_ptr(),
            tok_meta.data_ptr(), table_ptr, ctypes.byref(desc), out_split | (0x100 if qkv_f16 else 0), _stream()),
            'hfl_window_attention_fwd')
    return out


def window_attention_multi(problems, out_split: int = 2):
    """Several window-attention problems on fp16 (hi, lo) qkv operands as ONE launch when their shapes agree
    (hfl_window_attention_fwd_multi), else one launch each.  `problems`: list of dicts with the arguments of
    `window_attention` (qkv, tok_meta, rpe_table, n_tokens, n_windows, patch_size, dilation, n_relay, n_heads, batch_size,
    rt_row0, depth).  Returns the list of outputs (split2 bf16 rows for out_split 2, fp32 rows for 0)."""
    assert 1 <= len(problems) <= 4 and out_split in (0, 2)
    lib = _native.load()
    outs, descs, keep = [], [], []
    for pr in problems:
        qkv = _f32c(pr['qkv'])
        _dev(qkv, pr['tok_meta'], pr['rpe_table'])
        rows, c = qkv.shape[0], pr['n_heads'] * 16
        assert qkv.shape[1] == 3 * c
        out = (torch.zeros((rows, 2 * c), dtype=torch.bfloat16, device=qkv.device) if out_split == 2
               else torch.zeros((rows, c), dtype=torch.float32, device=qkv.device))
        bnd = int(0.8 * pr['patch_size'] * pr['dilation'] ** 0.5)
        desc = WindowAttnDesc(n_tokens=pr['n_tokens'], rt_row0=pr.get('rt_row0', 0), n_windows=pr['n_windows'],
                              patch_size=pr['patch_size'], dilation=pr['dilation'], n_relay=pr['n_relay'],
                              n_heads=pr['n_heads'], pos_bnd=bnd, batch_size=pr['batch_size'], scale=16 ** -0.5,
                              depth=pr['depth'])
        table = pr['rpe_table']
        if table is not None:
            expanded = rpe_expand(table, pr['n_heads'], bnd, pr['depth'], True)
            desc.rpe_expanded = None if expanded is None else expanded.data_ptr()
            table = _f32c(table)
            keep.append((expanded, table))
        outs.append(out)
        descs.append(desc)
        keep.append(qkv)
        pr['_ptrs'] = (out.data_ptr(), qkv.data_ptr(), pr['tok_meta'].data_ptr(), None if table is None else table.data_ptr())
    arr = lambda i: _native.ptr_array([pr['_ptrs'][i] for pr in problems])
    check(lib.hfl_window_attention_fwd_multi(len(problems), arr(0), arr(1), arr(2), arr(3),
                                             _native.ptr_array([ctypes.addressof(d) for d in descs]),
                                             out_split | 0x100, _stream()), 'hfl_window_attention_fwd_multi')
    return outs


def relay_attention_f16(qkv_f16, seq_rows, seq_off, batch: int, n_heads: int, max_seq_len: int, orphan_rows=None):
    """Ragged relay-token self-attention on the fp16 (hi, lo) operand rows of `ln_qkv_fused` (q pre-scaled by 16^-0.5 log2 e)
    -> (rows, 2C) bf16 split2, the operand of attention.proj; rows of no sequence (`orphan_rows`) come out zero
    (hfl_relay_attention_f16_fwd)."""
    _dev(qkv_f16, seq_rows, seq_off, orphan_rows)
    rows, c3 = qkv_f16.shape
    c = c3 // 3
    assert qkv_f16.dtype == torch.float32 and qkv_f16.is_contiguous() and c == n_heads * 16
    out = torch.empty((rows, 2 * c), dtype=torch.bfloat16, device=qkv_f16.device)
    n_orph = 0 if orphan_rows is None else orphan_rows.numel()
    check(_native.load().hfl_relay_attention_f16_fwd(out.data_ptr(), qkv_f16.data_ptr(), seq_rows.data_ptr(), seq_off.data_ptr(),
                                                     batch, n_heads, max_seq_len,
                                                     orphan_rows.data_ptr() if n_orph else None, n_orph, _stream()),
          'hfl_relay_attention_f16_fwd')
    return out


def relay_attention(qkv, seq_rows, seq_off, batch: int, n_heads: int, max_seq_len: int):
    """Ragged per-cloud attention over relay-token rows; rows in no sequence -> 0."""
    _dev(qkv, seq_rows, seq_off)
    qkv = _f32c(qkv)
    c = n_heads * 16
    out = torch.zeros((qkv.shape[0], c), dtype=torch.float32, device=qkv.device)
    with _timed('hfl_relay_attention_fwd', qkv.shape[0] * c * 16):
        check(_native.load().hfl_relay_attention_fwd(out.data_ptr(), qkv.data_ptr(),
                                                     seq_rows.data_ptr(), seq_off.data_ptr(), batch,
                                                     n_heads, 16 ** -0.5, int(max_seq_len), _stream()),
              'hfl_relay_attention_fwd')
    return out


def relay_token_init(x, tok_meta, n_windows: int, patch_size: int):
    _dev(x, tok_meta)
    x = _f32c(x)
    rt = torch.empty((n_windows, x.shape[1]), dtype=torch.float32, device=x.device)
    check(_native.load().hfl_relay_token_init(rt.data_ptr(), x.data_ptr(), tok_meta.data_ptr(),
                                              x.shape[0], n_windows, patch_size, x.shape[1],
                                              _stream()), 'hfl_relay_token_init')
    return rt


def window_stats(tok_meta, n_tokens: int, n_windows: int, patch_size: int, depth: int):
    _dev(tok_meta)
    stats = torch.empty((n_windows, 9), dtype=torch.float32, device=tok_meta.device)
    check(_native.load().hfl_window_stats(stats.data_ptr(), tok_meta.data_ptr(), n_tokens, n_windows,
                                          patch_size, depth, _stream()), 'hfl_window_stats')
    return stats


def segment_softmax_(scores, row_off, batch: int, scale: float):
    """In place: softmax over each cloud's rows, per query column."""
    _dev(scores, row_off)
    assert scores.dtype == torch.float32 and scores.is_contiguous() and row_off.dtype == torch.int64
    check(_native.load().hfl_segment_softmax(scores.data_ptr(), row_off.data_ptr(), batch,
                                             scores.shape[1], float(scale), _stream()),
          'hfl_segment_softmax')
    return scores


# --------------------------------------------------------------------------- octree
def token_meta(nkeys, depth: int):
    _dev(nkeys)
    assert nkeys.dtype == torch.int64
    meta = torch.empty((nkeys.shape[0], 2), dtype=torch.int32, device=nkeys.device)
    check(_native.load().hfl_token_meta(meta.data_ptr(), nkeys.data_ptr(), nkeys.shape[0], depth,
                                        _stream()), 'hfl_token_meta')
    return meta


def octree_neigh(neigh_parent, nidx, children, nkeys, depth: int, full_depth: int):
    _dev(nidx, children, nkeys)
    nne = nkeys.shape[0]
    out = torch.empty((nne, 27), dtype=torch.int32, device=nkeys.device)
    check(_native.load().hfl_octree_neigh(
        out.data_ptr(), None if neigh_parent is None else neigh_parent.data_ptr(), nidx.data_ptr(),
        children.data_ptr(), nkeys.data_ptr(), nne, depth, full_depth, _stream()), 'hfl_octree_neigh')
    return out
