"""Restatement of the `ocnn.nn` layers on the HOTFormerLoc hot path (TEST
INFRASTRUCTURE, see oracle/__init__.py).

Call sites in the reference: `models/layers/octformer_layers.py:89-95` and
`models/octformer_backbone.py:470-475` (OctreeConv), `libs/dwconv/dwconv/nn.py:49-63`
(OctreeDWConv, whose CUDA kernels `libs/dwconv/csrc/dwconv.cu:24-85` are asserted
equal to this op by `libs/dwconv/test/test_octree_dwconv.py:44-68`),
`models/layers/pooling.py:29,76` (OctreeGlobalPool, unused by shipped cfgs).
"""

from typing import List

import torch


def _kernel_string(kernel_size: List[int]) -> str:
    ks = list(kernel_size)
    if len(ks) == 1:
        ks = ks * 3
    return ''.join(str(k) for k in ks)


def octree_gather(data: torch.Tensor, neigh: torch.Tensor) -> torch.Tensor:
    """(N,C),(M,K) -> (M,K,C) with zeros where neigh < 0 (ocnn's octree2col)."""
    buf = data.new_zeros(neigh.shape[0], neigh.shape[1], data.shape[1])
    valid = neigh >= 0
    buf[valid] = data[neigh[valid]]
    return buf


def octree_pad(data: torch.Tensor, octree, depth: int, val: float = 0.0):
    mask = octree.nempty_mask(depth)
    out = data.new_full((mask.numel(), data.shape[1]), val)
    out[mask] = data
    return out


def octree_depad(data: torch.Tensor, octree, depth: int):
    return data[octree.nempty_mask(depth)]


class OctreeConv(torch.nn.Module):
    """weights (kdim, Cin, Cout) [+ bias (Cout)]; out = gather(data, neigh)
    .reshape(N, kdim*Cin) @ weights.reshape(kdim*Cin, Cout)."""

    def __init__(self, in_channels: int, out_channels: int,
                 kernel_size: List[int] = [3], stride: int = 1,
                 nempty: bool = False, direct_method: bool = False,
                 use_bias: bool = False, max_buffer: int = int(2e8)):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel = _kernel_string(kernel_size)
        self.kernel_size = [int(c) for c in self.kernel]
        self.kdim = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        self.stride = stride
        self.nempty = nempty
        self.use_bias = use_bias
        self.weights_shape = (self.kdim, in_channels, out_channels)
        self.weights = torch.nn.Parameter(torch.empty(*self.weights_shape))
        self.bias = torch.nn.Parameter(torch.empty(out_channels)) if use_bias else None
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.weights)
        if self.use_bias:
            torch.nn.init.zeros_(self.bias)

    def forward(self, data: torch.Tensor, octree, depth: int):
        neigh = octree.get_neigh(depth, self.kernel, self.stride, self.nempty)
        if not self.nempty and self.stride == 2:
            raise NotImplementedError('stride-2 conv on padded octrees is off the hot path')
        buf = octree_gather(data, neigh).flatten(1)
        out = buf @ self.weights.flatten(0, 1)
        if self.use_bias:
            out = out + self.bias
        return out


class OctreeDWConv(torch.nn.Module):
    """weights (kdim, 1, C); out[h,c] = sum_k w[k,0,c] * data[neigh[h,k], c]."""

    def __init__(self, in_channels: int, kernel_size: List[int] = [3],
                 stride: int = 1, nempty: bool = False, use_bias: bool = False,
                 **kwargs):
        super().__init__()
        self.in_channels = self.out_channels = in_channels
        self.kernel = _kernel_string(kernel_size)
        self.kernel_size = [int(c) for c in self.kernel]
        self.kdim = self.kernel_size[0] * self.kernel_size[1] * self.kernel_size[2]
        self.stride = stride
        self.nempty = nempty
        self.use_bias = use_bias
        self.weights = torch.nn.Parameter(torch.empty(self.kdim, 1, in_channels))
        self.bias = torch.nn.Parameter(torch.empty(in_channels)) if use_bias else None
        torch.nn.init.xavier_uniform_(self.weights)
        if use_bias:
            torch.nn.init.zeros_(self.bias)

    def forward(self, data: torch.Tensor, octree, depth: int):
        neigh = octree.get_neigh(depth, self.kernel, self.stride, self.nempty)
        buf = octree_gather(data, neigh)
        out = torch.einsum('ikc,kc->ic', buf, self.weights.flatten(0, 1))
        if self.use_bias:
            out = out + self.bias
        return out


class OctreeDeconv(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError('OctreeDeconv is not on the HOTFormerLoc hot path')


class OctreeGlobalPool(torch.nn.Module):
    """Per-cloud mean of node features via batch_id (count clamped >= 1)."""

    def __init__(self, nempty: bool = False):
        super().__init__()
        self.nempty = nempty

    def forward(self, data: torch.Tensor, octree, depth: int):
        bid = octree.batch_id(depth, self.nempty)
        B = octree.batch_size
        s = data.new_zeros(B, data.shape[1]).index_add_(0, bid, data)
        n = data.new_zeros(B).index_add_(0, bid, data.new_ones(bid.shape[0]))
        return s / n.clamp(min=1).unsqueeze(1)
