"""Drop-in for the reference's `dwconv` package (`libs/dwconv/dwconv/{__init__,nn}.py`) and
its compiled `dwconv.core` (`libs/dwconv/csrc/pybind.cpp:10-14`): same names, argument
meaning and autograd behaviour, HIP kernels underneath (csrc/dwconv.hip)."""

from typing import List

import torch
from torch.autograd import Function

from . import ops
from .ops import dwconv_forward_backward, dwconv_weight_backward, inverse_neigh  # noqa: F401 (dwconv.core)

__all__ = ['octree_dwconv', 'OctreeDWConv', 'dwconv_forward_backward', 'dwconv_weight_backward',
           'inverse_neigh']


class OctreeDWConvFunction(Function):
    """libs/dwconv/dwconv/nn.py:17-43"""

    @staticmethod
    def forward(ctx, data: torch.Tensor, weights: torch.Tensor, neigh: torch.Tensor):
        data, weights, neigh = data.contiguous(), weights.contiguous(), neigh.contiguous()
        out = ops.dwconv_forward_backward(data, weights, neigh)
        ctx.save_for_backward(data, weights, neigh)
        return out

    @staticmethod
    def backward(ctx, grad):
        data, weights, neigh = ctx.saved_tensors
        grad = grad.contiguous()
        grad_d = grad_w = None
        if ctx.needs_input_grad[0]:
            grad_d = ops.dwconv_forward_backward(grad, weights, ops.inverse_neigh(neigh))
        if ctx.needs_input_grad[1]:
            grad_w = ops.dwconv_weight_backward(grad, data, neigh)
        return grad_d, grad_w, None


octree_dwconv = OctreeDWConvFunction.apply


class OctreeDWConv(torch.nn.Module):
    """libs/dwconv/dwconv/nn.py:49-63 (parameter `weights` (kdim,1,C), optional `bias`)."""

    def __init__(self, channels: int, kernel_size: List[int] = [3], nempty: bool = False,
                 use_bias: bool = False):
        super().__init__()
        ks = list(kernel_size) * 3 if len(kernel_size) == 1 else list(kernel_size)
        self.kernel = ''.join(str(k) for k in ks)
        self.kdim = ks[0] * ks[1] * ks[2]
        self.stride = 1
        self.nempty = nempty
        self.use_bias = use_bias
        self.in_channels = self.out_channels = channels
        self.weights = torch.nn.Parameter(torch.empty(self.kdim, 1, channels))
        torch.nn.init.xavier_uniform_(self.weights)
        if use_bias:
            self.bias = torch.nn.Parameter(torch.zeros(channels))

    def forward(self, data: torch.Tensor, octree, depth: int):
        neigh = octree.get_neigh(depth, self.kernel, self.stride, self.nempty)
        out = octree_dwconv(data, self.weights, neigh)
        if self.use_bias:
            out = out + self.bias
        return out
