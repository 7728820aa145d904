"""Window / relay-token bookkeeping: the MI355X counterpart of `OctreeT`
(`models/octree.py:38-433` of the reference).

The reference materialises, per depth, int64 masks (N,K,K), relative positions
(N,K,K,3), padded batch-id streams and a (B,Rmax,Rmax) relay-token mask, and drives
ragged splits with `.tolist()` loops inside the forward.  Here the only things kept are

  * host integers derived from `batch_nnum_nempty` (already on the host after the
    octree build): token/window counts, per-cloud relay-token runs -- `window_layout`,
    pure numpy, testable without a GPU;
  * per depth one (N_t,2) uint32 token-meta array [x|y<<10|z<<20, batch id] on the
    device, from which the attention kernels derive masks and relative positions;
  * two small int32 index arrays describing each cloud's relay-token sequence.

The forward pass therefore contains no device->host synchronisation.
Formulas: SURVEY.md Appendix E (restating `models/octree.py:73-75,130-184,229-265`).
"""

from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops


def window_layout(batch_nne: np.ndarray, patch_size: int, dilation: int, max_depth: int,
                  start_depth: int, pyramid_depths: List[int]) -> dict:
    """Host-side integers of the window structure.

    batch_nne: (depth+1, B) non-empty node counts per depth and cloud.
    Returns per depth d in [start_depth, max_depth]:
        n_tokens[d], n_padded[d] (multiple of K*D, octree.py:73-75), n_windows[d] = n_padded/K,
      and per pyramid depth:
        num_windows[d] (B,)  windows owned by each cloud, straddling windows counted for the
                             left cloud, padding windows for the last (octree.py:163-184)
        first_window[d] (B,) index of each cloud's first window
        n_pad_windows[d]     windows that contain no real token
      plus the relay-token sequences of RTSA (relay_token_utils.py:25-39, octree.py:229-265):
        rt_offset[d]         first row of depth d inside the concatenated relay-token matrix
        seq_rows, seq_off    cloud b attends rows seq_rows[seq_off[b]:seq_off[b+1]]
        rt_counts (B,)       R_b including the padding windows of the last cloud (reference
                             `batch_num_relay_tokens_combined`)
    """
    K, D = patch_size, dilation
    B = batch_nne.shape[1]
    block = K * D
    out = dict(n_tokens={}, n_padded={}, n_windows={}, num_windows={}, first_window={},
               n_pad_windows={}, rt_offset={})
    for d in range(start_depth, max_depth + 1):
        nt = int(batch_nne[d].sum())
        na = -(-nt // block) * block
        out['n_tokens'][d] = nt
        out['n_padded'][d] = na
        out['n_windows'][d] = na // K
    row = 0
    for d in pyramid_depths:
        nt, na, W = out['n_tokens'][d], out['n_padded'][d], out['n_windows'][d]
        cum = np.cumsum(batch_nne[d].astype(np.int64))
        cum[-1] += na - nt
        boundary = cum // K + (cum % K != 0)
        nwin = np.diff(boundary, prepend=0)
        out['num_windows'][d] = nwin
        out['first_window'][d] = boundary - nwin
        out['n_pad_windows'][d] = W - (-(-nt // K))
        out['rt_offset'][d] = row
        row += W
    out['rt_rows_total'] = row
    seq_rows, seq_off = [], [0]
    rt_counts = np.zeros(B, dtype=np.int64)
    for b in range(B):
        for d in pyramid_depths:
            n = int(out['num_windows'][d][b])
            rt_counts[b] += n
            if b == B - 1:
                n -= out['n_pad_windows'][d]        # padding windows attend nothing (id B)
            start = out['rt_offset'][d] + int(out['first_window'][d][b])
            seq_rows.extend(range(start, start + max(n, 0)))
        seq_off.append(len(seq_rows))
    out['seq_rows'] = np.asarray(seq_rows, dtype=np.int32)
    out['seq_off'] = np.asarray(seq_off, dtype=np.int32)
    # rows of the relay-token matrix that no cloud lists: the relay tokens of the pure padding windows at the end of every depth
    orphans = []
    for d in pyramid_depths:
        W, npad = out['n_windows'][d], out['n_pad_windows'][d]
        orphans.extend(range(out['rt_offset'][d] + W - npad, out['rt_offset'][d] + W))
    out['orphan_rows'] = np.asarray(orphans, dtype=np.int32)
    out['rt_counts'] = rt_counts
    return out


class WindowPlan:
    """Device-side view of one batch octree for the attention stages."""

    def __init__(self, octree, patch_size: int, dilation: int, max_depth: int, start_depth: int,
                 num_pyramid_levels: int, num_octf_levels: int, adape_mode: Optional[str] = None):
        assert start_depth >= 1, 'Octree not deep enough for model depth'    # octree.py:71
        # strong reference: a checkpointed backward re-runs blocks after the caller's batch dict is gone.  The octree
        # caches its plans (WindowPlan.for_octree), so octree <-> plan is a reference cycle the garbage collector frees.
        self.octree = octree
        self.K, self.D = patch_size, dilation
        self.B = octree.batch_size
        self.max_depth, self.start_depth = max_depth, start_depth
        self.pyramid_depths = [max_depth - num_octf_levels - j for j in range(num_pyramid_levels)]
        self.adape_mode = adape_mode
        self.device = octree.device
        lay = window_layout(octree.batch_nnum_nempty.numpy(), patch_size, dilation, max_depth,
                            start_depth, self.pyramid_depths)
        self.layout = lay
        self.n_tokens: Dict[int, int] = lay['n_tokens']
        self.n_windows: Dict[int, int] = lay['n_windows']
        self.rt_offset: Dict[int, int] = lay['rt_offset']
        dev = self.device
        self.meta = {d: ops.token_meta(octree.nkeys[d], d) for d in range(start_depth, max_depth + 1)}
        # the small host-built index arrays travel in two copies (int32 | int64), not one blocking pageable copy each
        i32 = torch.from_numpy(np.concatenate([lay['seq_rows'], lay['seq_off'], lay['orphan_rows']])).to(dev, non_blocking=True)
        n1, n2 = lay['seq_rows'].size, lay['seq_rows'].size + lay['seq_off'].size
        self.seq_rows, self.seq_off, self.orphan_rows = i32[:n1], i32[n1:n2], i32[n2:]
        self.max_seq_len = int(np.diff(lay['seq_off']).max()) if self.B > 0 else 0
        # per-cloud row offsets of the token stream (attentional pooling segments) and the padded gather index of the
        # pooling head (built on the device: hfl_pad_index)
        self.cloud_off = {}
        self.pad_index = {}
        nne = octree.batch_nnum_nempty.numpy().astype(np.int64)
        offs = [np.concatenate([[0], np.cumsum(nne[d])]) for d in self.pyramid_depths]
        i64 = torch.from_numpy(np.concatenate(offs)).to(dev, non_blocking=True) if offs else None
        for j, d in enumerate(self.pyramid_depths):
            self.cloud_off[d] = i64[j * (self.B + 1):(j + 1) * (self.B + 1)]
            self.pad_index[d] = ops.pad_index(self.cloud_off[d], self.B, int(nne[d].max()))
        self.window_stats = {}
        if adape_mode is not None:
            if adape_mode != 'cov':
                raise NotImplementedError('ADaPE mode %r (shipped configs use "cov")' % adape_mode)
            for d in self.pyramid_depths:
                self.window_stats[d] = ops.window_stats(self.meta[d], self.n_tokens[d],
                                                        self.n_windows[d], self.K, d)

    @classmethod
    def for_octree(cls, octree, patch_size, dilation, max_depth, start_depth, num_pyramid_levels,
                   num_octf_levels, adape_mode=None):
        """The plan depends only on the octree and the model configuration: built once per (octree, cfg) and kept
        on the octree next to its neighbour tables (the reference rebuilds `OctreeT` in every forward,
        `models/hotformerloc_backbone.py:712-716`)."""
        key = (patch_size, dilation, max_depth, start_depth, num_pyramid_levels, num_octf_levels, adape_mode)
        cache = octree.__dict__.setdefault('_window_plans', {})
        plan = cache.get(key)
        if plan is None:
            plan = cls(octree, patch_size, dilation, max_depth, start_depth, num_pyramid_levels,
                       num_octf_levels, adape_mode)
            cache[key] = plan
        return plan

    def pad_index_for(self, depth: int):
        """(B * Nmax) gather index of a depth's rows, cloud by cloud, sentinel = one-past-the-end (a zero row)."""
        if depth not in self.pad_index:
            nne = self.octree.batch_nnum_nempty.numpy().astype(np.int64)[depth]
            off = torch.from_numpy(np.concatenate([[0], np.cumsum(nne)])).to(self.device)
            self.pad_index[depth] = ops.pad_index(off, self.B, int(nne.max()))
        return self.pad_index[depth]

    def cloud_count(self, depth: int):
        """(B,) number of rows of every cloud at `depth`, on the device."""
        cache = self.__dict__.setdefault('_cloud_count', {})
        if depth not in cache:
            cache[depth] = self.octree.batch_nnum_nempty[depth].to(self.device)
        return cache[depth]

    def row_cloud(self, depth: int, with_relay: bool):
        """int64 cloud index of every row of a depth's buffer: tokens by their batch id, relay rows
        (if any) by their window owner, padding windows -> last cloud (octformer_layers.py:262-281)."""
        key = (depth, with_relay)
        cache = self.__dict__.setdefault('_row_cloud', {})
        if key not in cache:
            tok = self.meta[depth][:, 1].long()
            if with_relay:
                first = torch.arange(self.n_windows[depth], device=self.device) * self.K
                own = torch.where(first < tok.shape[0], tok[first.clamp(max=tok.shape[0] - 1)],
                                  torch.full_like(first, self.B))
                tok = torch.cat([tok, own.clamp(max=self.B - 1)])
            cache[key] = tok
        return cache[key]

    def token_window(self, depth: int):
        """(win (n_tokens,) int64, keep (n_tokens, 1) float): the window of every token (dilation 1) and 1 where the token
        belongs to the window's owner cloud, 0 elsewhere -- the reference's `~rt_init_mask` (models/octree.py:143-145;
        the owner is the smallest batch id of the window = that of its first token)."""
        cache = self.__dict__.setdefault('_token_window', {})
        if depth not in cache:
            nt = self.n_tokens[depth]
            tok = self.meta[depth][:nt, 1].long()
            win = torch.arange(nt, device=self.device) // self.K
            keep = (tok == tok[win * self.K]).to(torch.float32).unsqueeze(1)
            cache[depth] = (win, keep)
        return cache[depth]

    def relay_cloud(self):
        """cloud index of every row of the concatenated relay-token matrix (all pyramid depths)."""
        if getattr(self, '_relay_cloud', None) is None:
            parts = [self.row_cloud(d, True)[self.n_tokens[d]:] for d in self.pyramid_depths]
            self._relay_cloud = torch.cat(parts)
        return self._relay_cloud

    def relay_pad_index(self):
        """(B, Rmax) row index of each cloud's relay-token sequence (-1 = padding) and its validity
        mask, for the dense training-path formulation of relay attention."""
        if getattr(self, '_relay_pad', None) is None:
            off = self.layout['seq_off']
            rmax = max(int(np.diff(off).max()), 1)
            idx = np.full((self.B, rmax), -1, dtype=np.int64)
            for b in range(self.B):
                rows = self.layout['seq_rows'][off[b]:off[b + 1]]
                idx[b, :len(rows)] = rows
            t = torch.from_numpy(idx).to(self.device)
            self._relay_pad = (t, t >= 0)
        return self._relay_pad

    def neigh(self, depth: int):
        return self.octree.get_neigh(depth, '333', 1, nempty=True)
