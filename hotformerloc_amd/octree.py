"""`Points` / `Octree` / `merge_octrees`: the ocnn-side API the reference's callers use,
re-designed for the MI355X path.

Reference call sites (the contract): `datasets/dataset_utils.py:89-94`
(`Points(cloud)`; `Octree(depth, full_depth).build_octree(points)`; `merge_octrees`),
`eval/pnv_evaluate.py:122-126,173-175`, `misc/torch_utils.py:47-51`
(`octree.to(device)`, `octree.construct_all_neigh()`), `models/octree.py:51-52`
(attributes copied into `OctreeT`).

Design (not ocnn's): an `Octree` here is *deferred*.  `build_octree` only records the
cloud; dataloader workers therefore ship raw points (48 KB per 4096-point cloud, not
a 7 MB octree).  The structure is materialised for the whole batch by ONE pair of HIP
launches (one workgroup per cloud, LDS bitonic sort -- csrc/octree.hip) the first time
the octree is on the GPU: `merge_octrees([...]).to('cuda')`, or immediately when the
points already live there.  The result is laid out for the kernels: int32 child and
neighbour tables over the *non-empty* nodes only, int64 keys kept for the API.
There is no CPU construction path: materialising without a GPU raises.
"""

from typing import List, Optional, Union

import numpy as np
import torch

from . import _native, ops
from ._native import check

_KERNEL_LUT = {   # ocnn kernel-string -> columns of the 27-neighbourhood
    '333': list(range(27)),
    '222': [13, 14, 16, 17, 22, 23, 25, 26],
    '311': [4, 13, 22], '131': [10, 13, 16], '113': [12, 13, 14],
    '331': [1, 4, 7, 10, 13, 16, 19, 22, 25],
    '313': [3, 4, 5, 12, 13, 14, 21, 22, 23],
    '133': [9, 10, 11, 12, 13, 14, 15, 16, 17],
}


class Points:
    """`ocnn.octree.Points` as the reference uses it: `Points(tensor (n,3) in [-1,1])`."""

    def __init__(self, points: torch.Tensor, normals=None, features=None, labels=None,
                 batch_id=None, batch_size: int = 1):
        if normals is not None or features is not None:
            raise NotImplementedError('only input feature "P" is on the HOTFormerLoc path')
        self.points = points
        self.normals = normals
        self.features = features
        self.labels = labels
        self.batch_id = batch_id
        self.batch_size = batch_size
        self.device = points.device

    def to(self, device, non_blocking: bool = False):
        return Points(self.points.to(device, non_blocking=non_blocking), batch_size=self.batch_size)

    def cuda(self, non_blocking: bool = False):
        return self.to('cuda', non_blocking)

    def cpu(self):
        return self.to('cpu')


def _split_ranges(edges: np.ndarray, step: int):
    """Cut the consecutive ranges [edges[k], edges[k+1]) into pieces of at most `step`: (range id, first, length) of every
    piece, in order (no Python loop over the ranges: this runs once per convolution and fresh batch)."""
    cnt = np.diff(edges)
    n = (cnt + step - 1) // step
    rid = np.repeat(np.arange(cnt.size, dtype=np.int64), n)
    within = np.arange(int(n.sum()), dtype=np.int64) - np.repeat(np.cumsum(n) - n, n)
    first = edges[rid] + within * step
    return rid, first, np.minimum(step, edges[rid + 1] - first)


class Octree:
    def __init__(self, depth: int, full_depth: int = 2, batch_size: int = 1,
                 device: Union[torch.device, str] = 'cpu', **kwargs):
        if depth > _native.HFL_OCTREE_MAX_DEPTH:
            raise ValueError('octree depth %d > %d unsupported' % (depth, _native.HFL_OCTREE_MAX_DEPTH))
        self.depth = depth
        self.full_depth = full_depth
        self.batch_size = batch_size
        self.device = torch.device(device)
        self._clouds: List[torch.Tensor] = []
        self._built = False
        num = depth + 1
        self.keys = [None] * num          # int64 (nnum_d)      all nodes
        self.children = [None] * num      # int32 (nnum_d)      non-empty rank or -1
        self.nkeys = [None] * num         # int64 (nne_d)       non-empty nodes only
        self.nidx = [None] * num          # int32 (nne_d)       position among all nodes
        self.neighs = [None] * num        # int32 (nne_d, 27)   non-empty -> non-empty
        self.points = [None] * num        # float32 (nne_depth, 3) at [depth]
        self.features = [None] * num
        self.normals = [None] * num
        self.nnum = torch.zeros(num, dtype=torch.int32)
        self.nnum_nempty = torch.zeros(num, dtype=torch.int32)
        self.batch_nnum = torch.zeros(num, batch_size, dtype=torch.int32)
        self.batch_nnum_nempty = torch.zeros(num, batch_size, dtype=torch.int32)

    # ------------------------------------------------------------- construction
    def build_octree(self, point_cloud: Points):
        pts = point_cloud.points
        if pts.dim() != 2 or pts.shape[1] != 3:
            raise ValueError('points must be (n, 3)')
        if pts.shape[0] < 1:
            raise ValueError('empty point cloud')
        self._clouds = [pts.detach().to(torch.float32).contiguous()]
        self.batch_size = 1
        self.device = pts.device
        self._built = False
        if pts.is_cuda:
            self._materialise()
        return None

    def _materialise(self):
        """Run the batched device builder on the recorded clouds."""
        if self._built:
            return
        if self.device.type != 'cuda':
            raise _native.NativeLibraryError(
                'octree construction runs on the GPU (HIP kernels); move the octree to a '
                'cuda device first -- there is no CPU construction path')
        lib = _native.load()
        dev = self.device
        B, D, F = len(self._clouds), self.depth, self.full_depth
        sizes = [int(c.shape[0]) for c in self._clouds]
        if max(sizes) > _native.HFL_OCTREE_MAX_POINTS:
            raise _native.NativeLibraryError(
                'cloud with %d points exceeds HFL_OCTREE_MAX_POINTS=%d'
                % (max(sizes), _native.HFL_OCTREE_MAX_POINTS))
        P = sum(sizes)
        pts = torch.cat([c.to(dev, non_blocking=True) for c in self._clouds]) if B > 1 \
            else self._clouds[0].to(dev)
        off_host = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int64)
        off = off_host.to(dev, non_blocking=True)
        scratch = torch.empty(int(lib.hfl_octree_scratch_bytes(P, B, max(sizes), D)), dtype=torch.uint8, device=dev)
        leaf_pts = torch.empty((P, 3), dtype=torch.float32, device=dev)
        counts = torch.zeros((D + 1, B), dtype=torch.int32, device=dev)
        st = ops._stream()
        check(lib.hfl_octree_build_clouds(pts.data_ptr(), off.data_ptr(), B, P, max(sizes), D, F,
                                          scratch.data_ptr(), leaf_pts.data_ptr(),
                                          counts.data_ptr(), st), 'hfl_octree_build_clouds')
        cnt = counts.cpu()                                   # the one host sync of a batch build
        self.batch_nnum_nempty = cnt.clone()
        bn = torch.zeros_like(cnt)
        for d in range(D + 1):
            bn[d] = (8 ** d) if d <= F else cnt[d - 1] * 8
        self.batch_nnum = bn
        self.nnum_nempty = cnt.sum(1).to(torch.int32)
        self.nnum = bn.sum(1).to(torch.int32)
        cum = torch.zeros((D + 1, B + 1), dtype=torch.int32)
        cum[:, 1:] = torch.cumsum(cnt, dim=1)
        cum_dev = cum.to(dev, non_blocking=True)
        for d in range(D + 1):
            nn_, ne = int(self.nnum[d]), int(self.nnum_nempty[d])
            self.keys[d] = torch.empty(nn_, dtype=torch.int64, device=dev)
            self.children[d] = torch.full((nn_,), -1, dtype=torch.int32, device=dev)
            self.nkeys[d] = torch.empty(ne, dtype=torch.int64, device=dev)
            self.nidx[d] = torch.empty(ne, dtype=torch.int32, device=dev)
        self.points[D] = torch.empty((int(self.nnum_nempty[D]), 3), dtype=torch.float32, device=dev)
        tab = lambda lst: _native.ptr_array([t.data_ptr() for t in lst])
        check(lib.hfl_octree_merge(scratch.data_ptr(), off.data_ptr(), cum_dev.data_ptr(), B, D, F, P,
                                   tab(self.keys), tab(self.children), tab(self.nkeys),
                                   tab(self.nidx), leaf_pts.data_ptr(), self.points[D].data_ptr(),
                                   st), 'hfl_octree_merge')
        self.batch_size = B
        self._built = True

    def construct_all_neigh(self):
        """ocnn builds (nnum_d,27) tables over all nodes; this builds the tables of the
        NON-EMPTY nodes (what `get_neigh(..., nempty=True)` returns) for d >= 1, plus the live-tap lists of
        the octree convolutions (`sparse_taps`) for the depths below the input depth's 3x3x3 table."""
        self._need_built()
        fresh = False
        for d in range(1, self.depth + 1):
            if self.neighs[d] is None:
                fresh = True
                self.neighs[d] = ops.octree_neigh(self.neighs[d - 1] if d > self.full_depth else None,
                                                  self.nidx[d], self.children[d], self.nkeys[d],
                                                  d, self.full_depth)
        if fresh and self.device.type == 'cuda':
            # what the live-tap kernels (on a stream of their own, below) have to wait for: these tables, not the whole stream
            self.__dict__['_neigh_event'] = torch.cuda.current_stream(self.device).record_event()
        if (fresh or '_sparse_taps' not in self.__dict__) and self.device.type == 'cuda':
            lo = max(self.full_depth + 1, 3)
            keys = [(d, '333', 1) for d in range(lo, self.depth)] + [(d, '222', 2) for d in range(lo, self.depth + 1)]
            self._start_tap_lists(keys)          # asynchronous: the first sparse_taps() of the forward collects the counts

    _FORWARD_CACHES = ('_window_plans', '_sparse_taps', '_tap_tiles', '_sparse_taps_bwd', '_tap_edges_dev', '_taps_pending')

    def drop_forward_caches(self):
        """Forget everything `model(batch)` derives from the octree and keeps on it between calls: the window / relay-token
        plans (the reference rebuilds `OctreeT` in every forward, `models/hotformerloc_backbone.py:712-716`), the live-tap
        lists and the row-tile tables of the octree convolutions (ocnn's `octree2col` gathers per call).  What stays is the
        ocnn state the reference's `to_device` hands to the model (`misc/torch_utils.py:47-51`): keys, children and the
        27-neighbour tables.  `bench.py` calls this at the top of every timed step (the reference's model boundary)."""
        for name in self._FORWARD_CACHES:
            self.__dict__.pop(name, None)

    def _need_built(self):
        if not self._built:
            self._materialise()

    # ---------------------------------------------------------------- accessors
    def nempty_mask(self, depth: int):
        self._need_built()
        return self.children[depth] >= 0

    def key(self, depth: int, nempty: bool = False):
        self._need_built()
        return self.nkeys[depth] if nempty else self.keys[depth]

    def batch_id(self, depth: int, nempty: bool = False):
        return self.key(depth, nempty) >> 48

    def xyzb(self, depth: int, nempty: bool = False):
        key = self.key(depth, nempty)
        meta = ops.token_meta(key, depth) if nempty else None
        if meta is None:
            k = key & ((1 << 48) - 1)
            x = torch.zeros_like(k); y = torch.zeros_like(k); z = torch.zeros_like(k)
            for i in range(depth):
                x |= ((k >> (3 * i + 2)) & 1) << i
                y |= ((k >> (3 * i + 1)) & 1) << i
                z |= ((k >> (3 * i)) & 1) << i
            return x, y, z, key >> 48
        p = meta[:, 0].long()
        return p & 1023, (p >> 10) & 1023, (p >> 20) & 1023, meta[:, 1].long()

    def get_neigh(self, depth: int, kernel: str = '333', stride: int = 1, nempty: bool = False):
        """ocnn `get_neigh`; only the non-empty form is on the hot path (all shipped cfgs
        use nempty=True).  stride 1: (nne_d, K) table; stride 2 + '222': the eight children
        of every non-empty parent = `children[d].view(-1, 8)`."""
        self._need_built()
        if not nempty:
            raise NotImplementedError('tables over empty nodes are off the HOTFormerLoc path')
        if stride == 2:
            if kernel != '222':
                raise NotImplementedError('stride-2 octree conv is only used with kernel 2x2x2')
            return self.children[depth].view(-1, 8)
        if stride != 1:
            raise ValueError('unsupported stride %d' % stride)
        if self.neighs[depth] is None:
            self.construct_all_neigh()
        neigh = self.neighs[depth]
        if kernel == '333':
            return neigh
        cols = torch.tensor(_KERNEL_LUT[kernel], device=neigh.device)
        return neigh[:, cols].contiguous()

    def sparse_taps(self, depth: int, kernel: str = '333', stride: int = 1):
        """Tap-major list of the live (row, tap) pairs of the 27-neighbour table at `depth` (octree convolutions over
        surface-like clouds touch 4-6 of their 27 taps): returns
            src   (P, 1) int32  input row of every pair, pairs ordered by tap then by output row,
            slot  (rows, taps) int32  position of (row, tap) in that list, -1 where the neighbour is missing,
            edges list of taps+1 ints (host)  pairs of tap k are [edges[k], edges[k+1]).
        kernel '333' stride 1: the 27-neighbour table; kernel '222' stride 2: the eight children of every parent.
        Built by `construct_all_neigh()` for every depth at once (HIP kernels, hfl_tap_lists) with ONE device->host
        read of all the per-tap counts -- i.e. at the reference's `to_device` boundary (`misc/torch_utils.py:47-51`),
        not inside `model(batch)`."""
        cache = self.__dict__.setdefault('_sparse_taps', {})
        key = (depth, kernel, stride)
        if key not in cache:
            if self.__dict__.get('_taps_pending'):
                self._finish_tap_lists()
            if key not in cache:
                self._build_tap_lists([key])
        return cache[key]

    def tap_tiles(self, depth: int, kernel: str, stride: int, w_rows: int):
        """Row tiles of the grouped tap GEMM (ops.linear_x3_grouped) over the pair list of `sparse_taps`: (n, 3) int32
        {first pair, pairs (<= 128), tap * w_rows}; no tile straddles a tap.  Device-built from the tap edges, cached."""
        cache = self.__dict__.setdefault('_tap_tiles', {})
        key = (depth, kernel, stride, w_rows)
        if key not in cache:
            _, _, edges = self.sparse_taps(depth, kernel, stride)
            n_tiles = sum((edges[k + 1] - edges[k] + 127) // 128 for k in range(len(edges) - 1))
            # built on the device from the tap edges already there (hfl_tap_tiles): no host table, no blocking copy
            cache[key] = ops.tap_tiles(self._tap_edges_dev[(depth, kernel, stride)], n_tiles, len(edges) - 1, w_rows)
        return cache[key]

    def sparse_taps_bwd(self, depth: int, kernel: str = '333', stride: int = 1):
        """What the backward of a live-tap convolution needs besides `sparse_taps`: `rowof` (P, 1) int32, the output row
        of every pair (to gather the output gradient pair-major), and `inv_slot` (n_src, taps) int32, for every INPUT row
        the pairs that read it (tap k: the pair of the output row whose k-th neighbour it is), -1 where there is none --
        the input gradient is then the same fixed-order slot sum as the forward (no atomics); plus the (chunks, tap_chunk_off)
        tables of `ops.tap_wgrad`."""
        cache = self.__dict__.setdefault('_sparse_taps_bwd', {})
        key = (depth, kernel, stride)
        if key not in cache:
            src, slot, edges = self.sparse_taps(depth, kernel, stride)
            neigh = self.get_neigh(depth, kernel, stride, nempty=True).contiguous()
            n_src = int(self.nnum_nempty[depth])
            rows = torch.arange(slot.shape[0], dtype=torch.int32, device=slot.device).view(-1, 1).expand_as(slot)
            # (scatter with the missing pairs sent to a spare entry: boolean-mask indexing is a nonzero + a host round trip in
            #  the middle of the backward)
            npairs = int(edges[-1])
            rowof = torch.empty(npairs + 1, dtype=torch.int32, device=slot.device)
            rowof.scatter_(0, torch.where(slot >= 0, slot, torch.full_like(slot, npairs)).reshape(-1).long(),
                           rows.reshape(-1))
            inv = torch.empty((n_src, neigh.shape[1]), dtype=torch.int32, device=neigh.device)
            ops.inverse_table(inv, neigh)
            inv_slot = torch.where(inv >= 0, slot.gather(0, inv.clamp_min(0).long()), torch.full_like(inv, -1))
            # pair chunks of the weight-gradient kernel: <= 2048 pairs each, never across a tap boundary
            e = np.asarray(edges, dtype=np.int64)
            tap, a, rows = _split_ranges(e, 2048)
            chunks_np = np.stack([tap, a, a + rows], 1).astype(np.int32)
            tap_off = np.concatenate([[0], np.cumsum((np.diff(e) + 2047) // 2048)]).tolist()
            cache[key] = (rowof[:edges[-1]].view(-1, 1), inv_slot.contiguous(),
                          torch.from_numpy(chunks_np).to(slot.device),
                          torch.tensor(tap_off, dtype=torch.int32, device=slot.device))
        return cache[key]

    def _build_tap_lists(self, keys):
        """Tap lists for several (depth, kernel, stride) tables: all launches, then ONE device->host read of the per-tap
        counts.  The read is asynchronous (pinned buffer + event): `construct_all_neigh()` only starts it, the first
        `sparse_taps()` of the forward waits for it -- by then the host has built the window plan and issued the first
        layers, so the round trip is hidden."""
        self._start_tap_lists(keys)
        self._finish_tap_lists()

    def _start_tap_lists(self, keys):
        cache = self.__dict__.setdefault('_sparse_taps', {})
        pending = self.__dict__.get('_taps_pending')
        todo = [k for k in keys if k not in cache and not (pending and k in pending[0])]
        if not todo:
            return
        if pending:
            self._finish_tap_lists()
        # The lists depend on the neighbour tables only, and the host needs their per-tap counts (one read) before it can size
        # the convolutions' buffers.  They run on the caller's stream, behind whatever the GPU is still running.  (On a
        # high-priority stream of their own the counts arrive earlier and the host runs a whole forward ahead -- and the step got
        # slower, 19 ms against 13: the just-ahead host had been enforcing the intended launch order of the H-OSA iterations for
        # free.  Round 4, profiles/r04_x_ab_tap_stream_resident.log.)
        cur = torch.cuda.current_stream(self.device)
        st = cur
        nev = self.__dict__.get('_neigh_event')
        if nev is not None:
            st.wait_event(nev)
        else:
            st.wait_stream(cur)
        with torch.cuda.stream(st):
            tables = [self.get_neigh(d, kern, stride, nempty=True).contiguous() for d, kern, stride in todo]
            taps = [t.shape[1] for t in tables]
            edges_all = torch.empty(sum(n + 1 for n in taps), dtype=torch.int32, device=self.device)
            edge_views, off = [], 0
            for n in taps:
                edge_views.append(edges_all[off:off + n + 1])
                off += n + 1
            built = []
            for i in range(0, len(tables), 16):                      # every table of the batch in three launches
                built += ops.tap_lists_multi(tables[i:i + 16], edge_views[i:i + 16])
            host = torch.empty(edges_all.numel(), dtype=torch.int32, pin_memory=True)
            host.copy_(edges_all, non_blocking=True)
            ev = st.record_event()
        self.__dict__['_taps_pending'] = (todo, built, taps, host, ev, edges_all)

    def _finish_tap_lists(self):
        pending = self.__dict__.pop('_taps_pending', None)
        if pending is None:
            return
        todo, built, taps, host, ev, edges_all = pending
        ev.synchronize()                                                  # the one host read
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)                                                # consumers run on the caller's stream
        edges_all.record_stream(cur)
        for src, slot, _ in built:
            src.record_stream(cur)
            slot.record_stream(cur)
        host = host.tolist()
        cache = self.__dict__.setdefault('_sparse_taps', {})
        dev_edges = self.__dict__.setdefault('_tap_edges_dev', {})
        off = 0
        for key, (src, slot, _), n in zip(todo, built, taps):
            edges = host[off:off + n + 1]
            dev_edges[key] = edges_all[off:off + n + 1]
            off += n + 1
            cache[key] = (src[:edges[-1]].view(-1, 1), slot, edges)

    def get_input_feature(self, feature: str = 'P', nempty: bool = True):
        """`ocnn.modules.InputFeature('P', nempty=True)` (models/hotformerloc.py:28-31):
        leaf averages rescaled from [0, 2^depth] to [-1, 1]."""
        self._need_built()
        if feature.upper() != 'P' or not nempty:
            raise NotImplementedError('only InputFeature("P", nempty=True) is supported')
        return self.points[self.depth] * (2.0 ** (1 - self.depth)) - 1.0

    # ------------------------------------------------------------------- device
    def to(self, device: Union[torch.device, str], non_blocking: bool = False):
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self.device == device:
            if device.type == 'cuda':
                self._need_built()
            return self
        out = Octree(self.depth, self.full_depth, self.batch_size, device)
        out._clouds = [c.to(device, non_blocking=non_blocking) for c in self._clouds]
        if self._built:
            mv = lambda lst: [t.to(device, non_blocking=non_blocking) if isinstance(t, torch.Tensor)
                              else None for t in lst]
            for name in ('keys', 'children', 'nkeys', 'nidx', 'neighs', 'points'):
                setattr(out, name, mv(getattr(self, name)))
            for name in ('nnum', 'nnum_nempty', 'batch_nnum', 'batch_nnum_nempty'):
                setattr(out, name, getattr(self, name).clone())          # stay on the host
            out._built = True
        elif device.type == 'cuda':
            out._materialise()
        return out

    def cuda(self, non_blocking: bool = False):
        return self.to('cuda', non_blocking)

    def cpu(self):
        return self.to('cpu')


def merge_octrees(octrees: List[Octree]) -> Octree:
    """`ocnn.octree.merge_octrees`: batch = the clouds in list order.  Deferred octrees are
    merged by concatenating their recorded clouds; the batch structure (keys with
    `batch << 48`, child offsets, stacked `batch_nnum*`) comes out of the batched builder."""
    if not octrees:
        raise ValueError('empty octree list')
    first = octrees[0]
    out = Octree(first.depth, first.full_depth, batch_size=0, device=first.device)
    for o in octrees:
        if o.depth != first.depth or o.full_depth != first.full_depth:
            raise ValueError('octrees of different depth cannot be merged')
        if not o._clouds:
            raise ValueError('octree without a point cloud')
        out._clouds.extend(o._clouds)
    out.batch_size = len(out._clouds)
    out.batch_nnum = torch.zeros(first.depth + 1, out.batch_size, dtype=torch.int32)
    out.batch_nnum_nempty = torch.zeros(first.depth + 1, out.batch_size, dtype=torch.int32)
    if all(c.is_cuda for c in out._clouds):
        out._materialise()
    return out


def build_batch_octree(clouds, depth: int, full_depth: int = 2, device='cuda',
                       construct_neigh: bool = True) -> Octree:
    """One-call fast path: list of (n,3) float32 arrays/tensors -> batch octree on `device`."""
    octree = Octree(depth, full_depth, batch_size=len(clouds), device=device)
    octree._clouds = [torch.as_tensor(c, dtype=torch.float32).contiguous() for c in clouds]
    octree.device = torch.device(device)
    if octree.device.type == 'cuda' and octree.device.index is None:
        octree.device = torch.device('cuda', torch.cuda.current_device())
    octree._materialise()
    if construct_neigh:
        octree.construct_all_neigh()
    return octree
