"""Restatement of the `ocnn==2.2.2` octree subset used by the HOTFormerLoc hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  `ocnn` is an un-vendored pip
dependency of the reference (`requirements.txt:6`, github.com/octree-nn/ocnn-pytorch)
and is not installable here; this file restates its *published* behaviour
 with torch-CPU ops.  It is pinned bit-exactly by ocnn's own
golden vectors which the reference vendors under `libs/dwconv/test/data/`
(`octree/test_00{1..5}.npz`, `batch_45.npz`), see tests/test_oracle_ocnn.py.

Reference call sites that define what must exist here:
  * `datasets/dataset_utils.py:89-94`   Points -> Octree.build_octree -> merge_octrees
  * `eval/pnv_evaluate.py:122-126,173-175`
  * `misc/torch_utils.py:47-51`         Octree.to(device) ; construct_all_neigh()
  * `models/octree.py:51-52,95-110,132,273-275,298`  Octree attrs, get_neigh, batch_id,
                                         key, key2xyz, xyzb
  * `libs/dwconv/dwconv/nn.py:59`       get_neigh(depth, kernel, stride, nempty)
"""

from typing import List, Optional, Union

import torch


# --------------------------------------------------------------------------- keys
def xyz2key(x, y, z, b=None, depth: int = 16):
    """Interleave the low `depth` bits of x,y,z (x most significant inside each
    triple): key = sum_i x_i<<(3i+2) | y_i<<(3i+1) | z_i<<(3i); batch index at bit 48.
    Coordinates are masked to `depth` bits (out-of-range values wrap, as with
    ocnn's byte lookup tables)."""
    x, y, z = x.long(), y.long(), z.long()
    key = torch.zeros_like(x)
    for i in range(depth):
        key = key | (((x >> i) & 1) << (3 * i + 2)) \
                  | (((y >> i) & 1) << (3 * i + 1)) \
                  | (((z >> i) & 1) << (3 * i))
    if b is not None:
        if isinstance(b, torch.Tensor):
            b = b.long()
        key = key | (b << 48)
    return key


def key2xyz(key, depth: int = 16):
    """Inverse of :func:`xyz2key`; returns (x, y, z, b) as int64 tensors."""
    b = key >> 48
    k = key & ((1 << 48) - 1)
    x = torch.zeros_like(k)
    y = torch.zeros_like(k)
    z = torch.zeros_like(k)
    for i in range(depth):
        x = x | (((k >> (3 * i + 2)) & 1) << i)
        y = y | (((k >> (3 * i + 1)) & 1) << i)
        z = z | (((k >> (3 * i)) & 1) << i)
    return x, y, z, b


# ------------------------------------------------------------------------- points
class Points:
    """Minimal `ocnn.octree.Points` (reference use: `Points(tensor (n,3))`)."""

    def __init__(self, points: torch.Tensor, normals: Optional[torch.Tensor] = None,
                 features: Optional[torch.Tensor] = None,
                 labels: Optional[torch.Tensor] = None,
                 batch_id: Optional[torch.Tensor] = None, batch_size: int = 1):
        self.points = points
        self.normals = normals
        self.features = features
        self.labels = labels
        self.batch_id = batch_id
        self.batch_size = batch_size
        self.device = points.device

    def to(self, device, non_blocking: bool = False):
        mv = lambda t: None if t is None else t.to(device, non_blocking=non_blocking)
        out = Points(mv(self.points), mv(self.normals), mv(self.features),
                     mv(self.labels), mv(self.batch_id), self.batch_size)
        return out

    def cpu(self):
        return self.to('cpu')

    def cuda(self, non_blocking: bool = False):
        return self.to('cuda', non_blocking)


# ------------------------------------------------------------------------- octree
_KERNEL_LUT = {
    '222': [13, 14, 16, 17, 22, 23, 25, 26],
    '311': [4, 13, 22],
    '131': [10, 13, 16],
    '113': [12, 13, 14],
    '331': [1, 4, 7, 10, 13, 16, 19, 22, 25],
    '313': [3, 4, 5, 12, 13, 14, 21, 22, 23],
    '133': [9, 10, 11, 12, 13, 14, 15, 16, 17],
}


def _neigh_luts():
    """(8,27) tables for the parent-walk: for child octant c=(cx,cy,cz) and
    offset o=(dx,dy,dz): parent offset floor((c+d)/2) and octant (c+d) mod 2."""
    lut_parent = torch.zeros(8, 27, dtype=torch.long)
    lut_child = torch.zeros(8, 27, dtype=torch.long)
    for c in range(8):
        cx, cy, cz = (c >> 2) & 1, (c >> 1) & 1, c & 1
        for o in range(27):
            dx, dy, dz = o // 9 - 1, (o // 3) % 3 - 1, o % 3 - 1
            tx, ty, tz = cx + dx, cy + dy, cz + dz
            px, py, pz = tx // 2, ty // 2, tz // 2            # floor division: -1,0,1
            lut_parent[c, o] = (px + 1) * 9 + (py + 1) * 3 + (pz + 1)
            lut_child[c, o] = ((tx & 1) << 2) | ((ty & 1) << 1) | (tz & 1)
    return lut_parent, lut_child


class Octree:
    def __init__(self, depth: int, full_depth: int = 2, batch_size: int = 1,
                 device: Union[torch.device, str] = 'cpu', **kwargs):
        self.depth = depth
        self.full_depth = full_depth
        self.batch_size = batch_size
        self.device = torch.device(device) if isinstance(device, str) else device
        self.reset()

    def reset(self):
        num = self.depth + 1
        self.keys = [None] * num
        self.children = [None] * num
        self.neighs = [None] * num
        self.features = [None] * num
        self.normals = [None] * num
        self.points = [None] * num
        self.nnum = torch.zeros(num, dtype=torch.int32)
        self.nnum_nempty = torch.zeros(num, dtype=torch.int32)
        self.batch_nnum = torch.zeros(num, self.batch_size, dtype=torch.int32)
        self.batch_nnum_nempty = torch.zeros(num, self.batch_size, dtype=torch.int32)

    # -- accessors -----------------------------------------------------------
    def nempty_mask(self, depth: int):
        return self.children[depth] >= 0

    def key(self, depth: int, nempty: bool = False):
        key = self.keys[depth]
        if nempty:
            key = key[self.nempty_mask(depth)]
        return key

    def xyzb(self, depth: int, nempty: bool = False):
        return key2xyz(self.key(depth, nempty), depth)

    def batch_id(self, depth: int, nempty: bool = False):
        bid = self.keys[depth] >> 48
        if nempty:
            bid = bid[self.nempty_mask(depth)]
        return bid

    # -- construction --------------------------------------------------------
    def build_octree(self, point_cloud: Points):
        """points in [-1,1] -> p = (points+1)*2^(depth-1), truncate, key, unique;
        full layers 0..full_depth; per level parent-unique with 8-child blocks."""
        self.device = point_cloud.points.device
        dev = self.device
        depth, full_depth = self.depth, self.full_depth

        scale = 2 ** (depth - 1)
        points = (point_cloud.points + 1.0) * scale
        key = xyz2key(points[:, 0], points[:, 1], points[:, 2], None, depth)
        node_key, idx, counts = torch.unique(
            key, sorted=True, return_inverse=True, return_counts=True)

        # full layers
        for d in range(full_depth + 1):
            n = 1 << (3 * d)
            self.keys[d] = torch.arange(n, dtype=torch.long, device=dev)
            self.children[d] = torch.arange(n, dtype=torch.int32, device=dev)
            self.nnum[d] = n
            self.nnum_nempty[d] = n

        # sparse layers, bottom-up
        for d in range(depth, full_depth, -1):
            pkey = node_key >> 3
            pkey, pidx = torch.unique_consecutive(pkey, return_inverse=True)
            k = (pkey.unsqueeze(-1) << 3) + torch.arange(8, device=dev)
            self.keys[d] = k.reshape(-1)
            self.nnum[d] = k.numel()
            self.nnum_nempty[d] = node_key.numel()
            addr = (pidx << 3) | (node_key & 7)
            children = torch.full((k.numel(),), -1, dtype=torch.int32, device=dev)
            children[addr] = torch.arange(node_key.numel(), dtype=torch.int32, device=dev)
            self.children[d] = children
            node_key = pkey

        # full_depth layer: mark which of the 8^full_depth nodes are non-empty
        d = full_depth
        children = torch.full_like(self.children[d], -1)
        children[node_key] = torch.arange(node_key.numel(), dtype=torch.int32, device=dev)
        self.children[d] = children
        self.nnum_nempty[d] = node_key.numel()

        # per-leaf averages at the finest layer
        cnt = counts.unsqueeze(1).to(points.dtype)
        pts = torch.zeros(counts.numel(), 3, dtype=points.dtype, device=dev)
        pts.index_add_(0, idx, points)
        self.points[depth] = pts / cnt
        if point_cloud.normals is not None:
            nrm = torch.zeros(counts.numel(), point_cloud.normals.shape[1],
                              dtype=points.dtype, device=dev)
            nrm.index_add_(0, idx, point_cloud.normals)
            nrm = nrm / cnt
            self.normals[depth] = torch.nn.functional.normalize(nrm, dim=1)
        if point_cloud.features is not None:
            ft = torch.zeros(counts.numel(), point_cloud.features.shape[1],
                             dtype=points.dtype, device=dev)
            ft.index_add_(0, idx, point_cloud.features)
            self.features[depth] = ft / cnt

        self.batch_nnum = self.nnum.clone().unsqueeze(1)
        self.batch_nnum_nempty = self.nnum_nempty.clone().unsqueeze(1)
        return idx

    def construct_neigh(self, depth: int):
        dev = self.device
        if depth <= self.full_depth:
            n = 1 << (3 * depth)
            key = torch.arange(n, dtype=torch.long, device=dev)
            x, y, z, _ = key2xyz(key, depth)
            xyz = torch.stack([x, y, z], dim=-1)                       # (n,3)
            g = torch.arange(-1, 2, device=dev)
            grid = torch.stack(torch.meshgrid(g, g, g, indexing='ij'), -1).view(27, 3)
            xyz = (xyz.unsqueeze(1) + grid).view(-1, 3)                 # (n*27,3)
            neigh = xyz2key(xyz[:, 0], xyz[:, 1], xyz[:, 2], None, depth)
            bs = torch.arange(self.batch_size, dtype=torch.long, device=dev)
            neigh = neigh.unsqueeze(0) + bs.unsqueeze(1) * n            # (B, n*27)
            bound = 1 << depth
            invalid = ((xyz < 0) | (xyz >= bound)).any(1)
            neigh[:, invalid] = -1
            self.neighs[depth] = neigh.view(-1, 27)
        else:
            lut_parent, lut_child = _neigh_luts()
            lut_parent, lut_child = lut_parent.to(dev), lut_child.to(dev)
            child_p = self.children[depth - 1]
            neigh_p = self.neighs[depth - 1][child_p >= 0]              # (Np,27)
            neigh_p = neigh_p[:, lut_parent]                            # (Np,8,27)
            child_pn = child_p.long()[neigh_p.clamp(min=0)]             # (Np,8,27)
            invalid = (child_pn < 0) | (neigh_p < 0)
            neigh = child_pn * 8 + lut_child
            neigh[invalid] = -1
            self.neighs[depth] = neigh.view(-1, 27)

    def construct_all_neigh(self):
        for d in range(1, self.depth + 1):
            self.construct_neigh(d)

    def get_neigh(self, depth: int, kernel: str = '333', stride: int = 1,
                  nempty: bool = False):
        if stride == 1:
            neigh = self.neighs[depth]
        elif stride == 2:
            neigh = self.neighs[depth][::8].clone()
        else:
            raise ValueError('Unsupported stride {}'.format(stride))
        if nempty:
            child = self.children[depth]
            if stride == 1:
                neigh = neigh[child >= 0]
            valid = neigh >= 0
            neigh[valid] = child[neigh[valid]].long()
        if kernel == '333':
            return neigh
        if kernel in _KERNEL_LUT:
            return neigh[:, torch.tensor(_KERNEL_LUT[kernel], device=neigh.device)]
        raise ValueError('Unsupported kernel {}'.format(kernel))

    def get_input_feature(self, feature: str, nempty: bool = False):
        depth = self.depth
        feature = feature.upper()
        feats = []
        if 'N' in feature:
            feats.append(self.normals[depth])
        if 'L' in feature or 'D' in feature:
            local = self.points[depth].frac() - 0.5
        if 'D' in feature:
            # displacement along the normal, in units of half the cell diagonal
            # (sqrt(3)/2); factor pinned by the 'ND' column of the ocnn fixtures
            dis = (self.normals[depth] * local).sum(1, keepdim=True)
            feats.append(dis * (2.0 / 3 ** 0.5))
        if 'L' in feature:
            feats.append(local)
        if 'P' in feature:
            feats.append(self.points[depth] * (2 ** (1 - depth)) - 1.0)
        if 'F' in feature:
            feats.append(self.features[depth])
        out = torch.cat(feats, dim=1)
        if not nempty:
            mask = self.nempty_mask(depth)
            full = out.new_zeros(mask.numel(), out.shape[1])
            full[mask] = out
            out = full
        return out

    # -- device --------------------------------------------------------------
    def to(self, device: Union[torch.device, str], non_blocking: bool = False):
        if isinstance(device, str):
            device = torch.device(device)
        if self.device == device:
            return self
        octree = Octree(self.depth, self.full_depth, self.batch_size, device)

        def mv(lst):
            return [t.to(device, non_blocking=non_blocking)
                    if isinstance(t, torch.Tensor) else None for t in lst]
        octree.keys = mv(self.keys)
        octree.children = mv(self.children)
        octree.neighs = mv(self.neighs)
        octree.features = mv(self.features)
        octree.normals = mv(self.normals)
        octree.points = mv(self.points)
        octree.nnum = self.nnum.clone()                       # stay on CPU
        octree.nnum_nempty = self.nnum_nempty.clone()
        octree.batch_nnum = self.batch_nnum.clone()
        octree.batch_nnum_nempty = self.batch_nnum_nempty.clone()
        return octree

    def cuda(self, non_blocking: bool = False):
        return self.to('cuda', non_blocking)

    def cpu(self):
        return self.to('cpu')


def merge_octrees(octrees: List[Octree]) -> Octree:
    """Per depth concat in batch order; key |= i<<48; child offsets by the
    cumulative non-empty count; batch_nnum(_nempty) stacked (depth+1, B)."""
    out = Octree(octrees[0].depth, octrees[0].full_depth,
                 batch_size=len(octrees), device=octrees[0].device)
    B = len(octrees)
    out.batch_nnum = torch.stack([o.nnum for o in octrees], dim=1)
    out.batch_nnum_nempty = torch.stack([o.nnum_nempty for o in octrees], dim=1)
    out.nnum = out.batch_nnum.sum(1).to(torch.int32)
    out.nnum_nempty = out.batch_nnum_nempty.sum(1).to(torch.int32)
    cum = torch.cumsum(out.batch_nnum_nempty, dim=1)
    for d in range(out.depth + 1):
        keys, children = [], []
        for i, o in enumerate(octrees):
            keys.append(o.keys[d] | (i << 48))
            c = o.children[d].clone()
            if i > 0:
                c[c >= 0] += int(cum[d, i - 1])
            children.append(c)
        out.keys[d] = torch.cat(keys)
        out.children[d] = torch.cat(children)
        for name in ('features', 'normals', 'points'):
            parts = [getattr(o, name)[d] for o in octrees]
            if all(p is not None for p in parts):
                getattr(out, name)[d] = torch.cat(parts)
    return out
