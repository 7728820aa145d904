"""Drop-in for the reference's `dwconv` package (`libs/dwconv/dwconv/{__init__,nn}.py`) and
its compiled `dwconv.core` (`libs/dwconv/csrc/pybind.cpp:10-14`): the same public names
(`octree_dwconv`, `OctreeDWConvFunction`, `OctreeDWConv`, `dwconv_forward_backward`,
`dwconv_weight_backward`, `inverse_neigh`), argument meaning and autograd behaviour, on the HIP
kernels of csrc/dwconv.hip.

One difference in mechanism: the data gradient needs the inverse neighbour table; the reference rebuilds it
in every backward call (`nn.py:36-38`), here it is built once per neighbour table and kept while that
table is alive (the octree caches its tables, so all 37 CPE calls of a step share three of them)."""

import weakref
from typing import List, Optional, Sequence

import torch
from torch.autograd import Function

from . import ops
from .ops import dwconv_forward_backward, dwconv_weight_backward, inverse_neigh  # noqa: F401 (dwconv.core)

__all__ = ['octree_dwconv', 'OctreeDWConv', 'OctreeDWConvFunction', 'dwconv_forward_backward',
           'dwconv_weight_backward', 'inverse_neigh']

_INVERSE_TABLES = {}        # id(table) -> (weakref to the table, its version, inverse table)


def _inverse_of(table: torch.Tensor) -> torch.Tensor:
    key = id(table)
    hit = _INVERSE_TABLES.get(key)
    if hit is not None and hit[0]() is table and hit[1] == table._version:
        return hit[2]
    inv = ops.inverse_neigh(table)
    if hit is None:                       # the entry dies with its table (training sees a new octree every batch)
        weakref.finalize(table, _INVERSE_TABLES.pop, key, None)
    _INVERSE_TABLES[key] = (weakref.ref(table), table._version, inv)
    return inv


class OctreeDWConvFunction(Function):
    """out[h, c] = sum_k weights[k, 0, c] * data[neigh[h, k], c]  (libs/dwconv/dwconv/nn.py:17-43).
    d/d data is the same kernel driven by the inverse table, d/d weights a slab reduction
    (`hfl_dwconv_weight_backward`)."""

    @staticmethod
    def forward(ctx, data: torch.Tensor, weights: torch.Tensor, neigh: torch.Tensor):
        operands = tuple(t.contiguous() for t in (data, weights, neigh))
        ctx.save_for_backward(*operands)
        return ops.dwconv_forward_backward(*operands)

    @staticmethod
    def backward(ctx, grad_out):
        data, weights, neigh = ctx.saved_tensors
        want_data, want_weights = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g = grad_out.contiguous()
        d_data = ops.dwconv_forward_backward(g, weights, _inverse_of(neigh)) if want_data else None
        d_weights = ops.dwconv_weight_backward(g, data, neigh) if want_weights else None
        return d_data, d_weights, None


octree_dwconv = OctreeDWConvFunction.apply


def _kernel_code(kernel_size: Sequence[int]) -> str:
    ks = list(kernel_size) * 3 if len(kernel_size) == 1 else list(kernel_size)
    assert len(ks) == 3
    return ''.join(str(k) for k in ks)


class OctreeDWConv(torch.nn.Module):
    """`dwconv.OctreeDWConv(channels, kernel_size=[3], nempty=False, use_bias=False)`
    (libs/dwconv/dwconv/nn.py:49-63 over `ocnn.nn.OctreeDWConv`): parameter `weights` (kdim, 1, C),
    optional `bias` (C); `forward(data, octree, depth)`."""

    def __init__(self, channels: int, kernel_size: List[int] = [3], nempty: bool = False,
                 use_bias: bool = False):
        super().__init__()
        self.kernel = _kernel_code(kernel_size)
        self.kdim = int(self.kernel[0]) * int(self.kernel[1]) * int(self.kernel[2])
        self.in_channels = self.out_channels = channels
        self.stride, self.nempty, self.use_bias = 1, nempty, use_bias
        self.weights = torch.nn.Parameter(torch.empty(self.kdim, 1, channels))
        torch.nn.init.xavier_uniform_(self.weights)
        self.bias: Optional[torch.nn.Parameter] = torch.nn.Parameter(torch.zeros(channels)) if use_bias else None

    def forward(self, data: torch.Tensor, octree, depth: int):
        table = octree.get_neigh(depth, self.kernel, self.stride, self.nempty)
        y = octree_dwconv(data, self.weights, table)
        return y if self.bias is None else y + self.bias
