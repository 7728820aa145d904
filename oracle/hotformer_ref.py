"""CPU oracle for the HOTFormerLoc hot path: octree -> global descriptor.

TEST INFRASTRUCTURE (see oracle/__init__.py).  A functional, torch-CPU fp32
restatement of the reference forward.  It keeps the reference's *materialised*
formulation on purpose (padded token streams, additive -1e3 masks, gathered RPE
bias, padded relay-token batches) so that it can be read side by side with the
reference; the product (`hotformerloc_amd/`) computes the same function without
materialising any of it.  Pinned against the reference's own Python files run in
the build container (`oracle/gen_golden.py` -> `tests/golden/model_*.npz`).

Reference lines followed by each function are cited in its docstring
(paths relative to the reference root).
"""

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from oracle.ocnn_ref.octree import key2xyz
from oracle.ocnn_ref.nn import octree_gather

MASK_VALUE = -1e3          # models/octree.py:66


# --------------------------------------------------------------------- small ops
def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def _linear(x, sd, p):
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def _mlp(x, sd, p):
    """models/layers/octformer_layers.py:53-59 (dropout = identity in eval)."""
    return _linear(F.gelu(_linear(x, sd, p + '.fc1')), sd, p + '.fc2')


def _sdpa(q, k, v, bias, scale):
    """softmax(q k^T * scale + bias) v  -- F.scaled_dot_product_attention semantics
    (models/octformer_backbone.py:83-88, hotformerloc_backbone.py:99-101, salsa.py:39-41)."""
    s = torch.matmul(q, k.transpose(-2, -1)) * scale + bias
    return torch.matmul(torch.softmax(s, dim=-1), v)


def _pair_mask(ids: torch.Tensor) -> torch.Tensor:
    """models/octree.py:267-270: (.., L) ids -> (.., L, L) float {0, -1e3}."""
    d = ids.unsqueeze(-1) - ids.unsqueeze(-2)
    return (d != 0).to(torch.float32) * MASK_VALUE


def _pad_rows(seqs: List[torch.Tensor], fill=0):
    """models/octree.py:21-35 pad_sequence."""
    n = max(s.shape[0] for s in seqs)
    out = seqs[0].new_full((len(seqs), n) + tuple(seqs[0].shape[1:]), fill)
    for i, s in enumerate(seqs):
        out[i, :s.shape[0]] = s
    return out


# --------------------------------------------------------------- octree convs
def octree_conv(data, sd, p, octree, depth, kernel: str, stride: int):
    """ocnn.nn.OctreeConv: gather -> (N, kdim*Cin) @ (kdim*Cin, Cout)."""
    neigh = octree.get_neigh(depth, kernel, stride, nempty=True)
    out = octree_gather(data, neigh).flatten(1) @ sd[p + '.weights'].flatten(0, 1)
    if (p + '.bias') in sd:
        out = out + sd[p + '.bias']
    return out


def octree_dwconv(data, weights, neigh):
    """libs/dwconv/csrc/dwconv.cu:24-42 == ocnn.nn.OctreeDWConv:
    out[h,c] = sum_k [neigh[h,k]>=0] w[k,0,c] data[neigh[h,k],c]."""
    return torch.einsum('ikc,kc->ic', octree_gather(data, neigh), weights.flatten(0, 1))


def conv_norm_relu(data, sd, p, octree, depth, kernel, stride):
    """models/layers/octformer_layers.py:94-98."""
    return F.relu(_ln(octree_conv(data, sd, p + '.conv', octree, depth, kernel, stride),
                      sd, p + '.norm'))


def downsample(data, sd, p, octree, depth):
    """models/octformer_backbone.py:474-477."""
    return _ln(octree_conv(data, sd, p + '.conv', octree, depth, '222', 2), sd, p + '.norm')


def cpe(data, sd, p, octree, depth):
    """models/layers/octformer_layers.py:138-142 (xcpe=False -> linear is Identity)."""
    neigh = octree.get_neigh(depth, '333', 1, nempty=True)
    return _ln(octree_dwconv(data, sd[p + '.conv.weights'], neigh), sd, p + '.norm')


# ------------------------------------------------------------------ window plan
class WindowPlan:
    """Restatement of `OctreeT` (models/octree.py:44-93,112-344): everything the
    attention layers read, for depths start_depth..max_depth."""

    def __init__(self, octree, patch_size, dilation, max_depth, start_depth,
                 num_pyramid_levels, num_octf_levels, adape_mode=None):
        self.octree = octree
        self.K, self.D = patch_size, dilation
        self.B = octree.batch_size
        self.max_depth, self.start_depth = max_depth, start_depth
        self.pyramid_depths = [max_depth - num_octf_levels - j
                               for j in range(num_pyramid_levels)]
        self.adape_mode = adape_mode
        block = patch_size * dilation                              # octree.py:73-75
        self.nnum_t = octree.nnum_nempty.clone().long()
        self.nnum_a = ((self.nnum_t + block - 1) // block) * block
        n = max_depth + 1
        self.batch_idx = [None] * n
        self.rt_init_mask = [None] * n
        self.rt_batch_idx = [None] * n
        self.batch_num_windows = [None] * n
        self.patch_mask = [None] * n
        self.dilate_mask = [None] * n
        self.hat_mask = [None] * n
        self.rel_pos = [None] * n
        self.dilate_pos = [None] * n
        self.window_stats = [None] * n
        rt_layers = [False] * num_octf_levels + [True] * num_pyramid_levels
        for i, d in enumerate(range(start_depth, max_depth + 1)):   # octree.py:119-125
            self._build_depth(d, rt_layers[-(i + 1)])
        self._build_rt_mask()

    # -- padding / windows (octree.py:346-386) -------------------------------
    def pad(self, data, depth, fill=0):
        num = int(self.nnum_a[depth] - self.nnum_t[depth])
        tail = data.new_full((num,) + tuple(data.shape[1:]), fill)
        return torch.cat([data, tail], 0)

    def to_windows(self, data, depth, dilated, fill=0):
        C = data.shape[-1]
        data = self.pad(data, depth, fill)
        if dilated:
            data = data.view(-1, self.K, self.D, C).transpose(1, 2).reshape(-1, C)
        return data.view(-1, self.K, C)

    def from_windows(self, data, depth, dilated):
        C = data.shape[-1]
        data = data.reshape(-1, C)
        if dilated:
            data = data.view(-1, self.D, self.K, C).transpose(1, 2).reshape(-1, C)
        return data[:int(self.nnum_t[depth])]

    def _build_depth(self, d, use_rt):
        o, K, D, B = self.octree, self.K, self.D, self.B
        bid = self.pad(o.batch_id(d, nempty=True), d, B)            # octree.py:132-134
        self.batch_idx[d] = bid
        w = bid.view(-1, K)
        self.patch_mask[d] = _pair_mask(w)                          # octree.py:193-199
        self.dilate_mask[d] = _pair_mask(w.view(-1, K, D).transpose(1, 2).reshape(-1, K))
        x, y, z, _ = key2xyz(self.pad(o.key(d, nempty=True), d), d)  # octree.py:272-283
        xyz = torch.stack([x, y, z], 1).view(-1, K, 3)
        self.rel_pos[d] = xyz.unsqueeze(2) - xyz.unsqueeze(1)
        xd = xyz.view(-1, K, D, 3).transpose(1, 2).reshape(-1, K, 3)
        self.dilate_pos[d] = xd.unsqueeze(2) - xd.unsqueeze(1)
        if not use_rt:
            return
        owner = w.min(1, keepdim=True).values                      # octree.py:142-154
        self.rt_init_mask[d] = w != owner
        self.hat_mask[d] = _pair_mask(torch.cat([owner, w], 1))
        self.rt_batch_idx[d] = owner.squeeze(1)
        cum = o.batch_nnum_nempty[d].long().cumsum(0)              # octree.py:163-184
        cum[-1] += self.nnum_a[d] - self.nnum_t[d]
        boundary = cum // K + (cum % K != 0).long()
        self.batch_num_windows[d] = torch.diff(boundary, prepend=boundary.new_zeros(1))
        if self.adape_mode is not None:
            self.window_stats[d] = self._window_stats(d)

    def _window_stats(self, d):
        """octree.py:285-344 (mode 'cov'/'var'/'pos')."""
        o = self.octree
        x, y, z, _ = o.xyzb(d, nempty=True)
        pts = torch.stack([x, y, z], 1).float() * (2 ** (1 - d)) - 1.0   # misc/utils.py:293-304
        pts = self.to_windows(pts, d, dilated=False)
        valid = ~self.rt_init_mask[d]
        cnt = valid.sum(1, keepdim=True).float()
        mu = (pts * valid.unsqueeze(-1)).sum(1) / cnt.clamp(min=1.0)
        nfeat = {'pos': 3, 'var': 6, 'cov': 9}[self.adape_mode]
        stats = torch.zeros(pts.shape[0], nfeat)
        stats[:, :3] = mu
        if self.adape_mode in ('var', 'cov'):
            cen = (pts - mu.unsqueeze(1)) * valid.unsqueeze(-1)
            den = (cnt - 1).clamp(min=1.0)
            if self.adape_mode == 'var':
                stats[:, 3:] = (cen ** 2).sum(1) / den * (cnt >= 2).float()
            else:
                cov = torch.bmm(cen.transpose(1, 2), cen) / den.unsqueeze(-1)
                cov = cov * (cnt >= 2).unsqueeze(-1).float()
                iu = torch.triu_indices(3, 3)
                stats[:, 3:] = cov[:, iu[0], iu[1]]
        return stats

    def _build_rt_mask(self):
        """octree.py:229-265."""
        B = self.B
        self.rt_counts = sum(self.batch_num_windows[d] for d in self.pyramid_depths)
        R = int(self.rt_counts.max())
        ids = torch.full((B, R), 10000, dtype=torch.long)
        for b in range(B):
            ids[b, :int(self.rt_counts[b])] = b
        prev = 0
        for d in self.pyramid_depths:
            npad = int((self.rt_batch_idx[d] >= B).sum())
            last = int(self.batch_num_windows[d][-1])
            if npad > 0:
                ids[-1, prev + last - npad: prev + last] = B
            prev += last
        self.rt_attn_mask = _pair_mask(ids)


# -------------------------------------------------------------------- attention
def rpe_bias(table, rel_pos, patch_size, dilation):
    """models/layers/octformer_layers.py:156-170; table (3*(2*bnd+1), H)."""
    bnd = int(0.8 * patch_size * dilation ** 0.5)
    n = 2 * bnd + 1
    idx = rel_pos.clamp(-bnd, bnd) + (bnd + torch.arange(3) * n)
    out = table.index_select(0, idx.reshape(-1)).view(idx.shape + (-1,)).sum(3)
    return out.permute(0, 3, 1, 2)                                   # (N,H,K,K)


def window_attention(x, sd, p, mask, rel_pos, H, K, G, dilation):
    """models/octformer_backbone.py:52-93."""
    C = x.shape[-1]
    qkv = _linear(x, sd, p + '.qkv').reshape(-1, K + G, 3, H, C // H).permute(2, 0, 3, 1, 4)
    bias = mask.unsqueeze(1)
    rpe = rpe_bias(sd[p + '.rpe.rpe_table'], rel_pos, K, dilation)
    if G > 0:
        rpe = F.pad(rpe, (G, 0, G, 0))
    bias = bias + rpe
    out = _sdpa(qkv[0], qkv[1], qkv[2], bias, (C // H) ** -0.5)
    out = out.transpose(1, 2).reshape(-1, K + G, C)
    return _linear(out, sd, p + '.proj')


def octformer_block(x, sd, p, plan, depth, H, dilation):
    """models/octformer_backbone.py:251-299 (use_rt=False)."""
    K = plan.K
    dil = dilation > 1
    x = x + cpe(x, sd, p + '.cpe', plan.octree, depth)
    x = plan.to_windows(x, depth, dil)
    mask = plan.dilate_mask[depth] if dil else plan.patch_mask[depth]
    pos = plan.dilate_pos[depth] if dil else plan.rel_pos[depth]
    x = x + window_attention(_ln(x, sd, p + '.norm1'), sd, p + '.attention', mask, pos,
                             H, K, 0, dilation)
    x = x + _mlp(_ln(x, sd, p + '.norm2'), sd, p + '.mlp')
    return plan.from_windows(x, depth, dil)


def hosa_block(x, rt, sd, p, plan, depth, H):
    """models/hotformerloc_backbone.py:197-236 (rt_propagation off)."""
    K = plan.K
    x = x + cpe(x, sd, p + '.cpe', plan.octree, depth)
    x = plan.to_windows(x, depth, False)
    x = torch.cat([rt.unsqueeze(1), x], 1)
    x = x + window_attention(_ln(x, sd, p + '.norm1'), sd, p + '.attention',
                             plan.hat_mask[depth], plan.rel_pos[depth], H, K, 1, 1)
    x = x + _mlp(_ln(x, sd, p + '.norm2'), sd, p + '.mlp')
    rt, x = x[:, 0], x[:, 1:]
    return plan.from_windows(x, depth, False), rt


def rtsa_block(rts: Dict[int, torch.Tensor], sd, p, plan, H):
    """models/hotformerloc_backbone.py:275-295,83-119; relay_token_utils.py:12-79."""
    depths = plan.pyramid_depths
    B = plan.B
    counts = [plan.batch_num_windows[d].tolist() for d in depths]
    split = [rts[d].split(counts[j]) for j, d in enumerate(depths)]
    x = _pad_rows([torch.cat([split[j][b] for j in range(len(depths))]) for b in range(B)])
    C = x.shape[-1]
    h = _ln(x, sd, p + '.norm1')
    qkv = _linear(h, sd, p + '.rt_attention.qkv').reshape(B, -1, 3, H, C // H).permute(2, 0, 3, 1, 4)
    a = _sdpa(qkv[0], qkv[1], qkv[2], plan.rt_attn_mask.unsqueeze(1), (C // H) ** -0.5)
    a = a.transpose(1, 2).reshape(B, -1, C)
    x = x + _linear(a, sd, p + '.rt_attention.proj')
    x = x + _mlp(_ln(x, sd, p + '.norm2'), sd, p + '.mlp')
    out = {d: [] for d in depths}
    for b in range(B):
        seq = x[b, :int(plan.rt_counts[b])]
        parts = seq.split([counts[j][b] for j in range(len(depths))])
        for j, d in enumerate(depths):
            out[d].append(parts[j])
    return {d: torch.cat(out[d]) for d in depths}


def relay_token_init(x, sd, p, plan, depth, use_cpe):
    """models/hotformerloc_backbone.py:345-363: masked window mean (nanmean)."""
    if use_cpe:
        x = cpe(x, sd, p + '.cpe', plan.octree, depth)
    x = plan.pad(x, depth).view(-1, plan.K, x.shape[-1])
    x = x.masked_fill(plan.rt_init_mask[depth].unsqueeze(-1), float('nan'))
    return torch.nanmean(x, dim=1)


# ---------------------------------------------------------------------- pooling
def pyramid_attn_pool_mixer(feats: Dict[int, torch.Tensor], sd, p, plan, k_tokens):
    """models/layers/pooling.py:183-233, salsa.py:25-55,103-111."""
    o = plan.octree
    B = plan.B
    toks = []
    for j, d in enumerate(feats.keys()):
        counts = o.batch_nnum_nempty[d].tolist()
        x = _pad_rows(list(feats[d].split(counts)))                       # (B,Nmax,C)
        ids = torch.full(x.shape[:2], 10000, dtype=torch.long)
        for b, n in enumerate(counts):
            ids[b, :n] = b
        mask = _pair_mask(ids)[:, 0, :].unsqueeze(1)                      # (B,1,Nmax)
        q = sd['%s.attpool.%d.query' % (p, j)].unsqueeze(0).expand(B, -1, -1)
        toks.append(_sdpa(q, x, x, mask, x.shape[-1] ** -0.5))
    x = torch.cat(toks, 1)                                                # (B, sum k, C)
    e = p + '.descriptor_extractor'
    i = 0
    while ('%s.mix.%d.mix.0.weight' % (e, i)) in sd:
        m = '%s.mix.%d.mix' % (e, i)
        x = x + _linear(F.gelu(_linear(_ln(x, sd, m + '.0'), sd, m + '.1')), sd, m + '.3')
        i += 1
    x = _linear(x.permute(0, 2, 1), sd, e + '.channel_proj').permute(0, 2, 1)
    x = _linear(x, sd, e + '.row_proj')
    return x.flatten(1)


# ---------------------------------------------------------------------- forward
def forward_with_grad(sd: Dict[str, torch.Tensor], params, octree, capture: Optional[dict] = None):
    """models/hotformerloc.py:33-59 -> hotformerloc_backbone.py:702-723,574-635.

    sd:      state_dict (CPU fp32) with the reference's key names
    params:  object with the reference `ModelParams` fields (misc/utils.py:15-115)
    octree:  merged `oracle.ocnn_ref.Octree` with `construct_all_neigh()` done
    capture: optional dict that receives named intermediates
    """
    cap = capture if capture is not None else {}
    depth = octree.depth
    nlev, noctf = params.num_pyramid_levels, params.num_octf_levels
    heads = list(params.num_heads) if params.num_heads else [c // 16 for c in params.channels]
    K, dil = params.patch_size, params.dilation
    adape = getattr(params, 'ADaPE_mode', None)
    stem_down = params.num_input_downsamples
    bb = 'backbone.backbone'

    # input feature 'P' (hotformerloc.py:28-31)
    x = octree.points[depth] * (2 ** (1 - depth)) - 1.0
    cap['input_feature'] = x

    # patch embed (octformer_backbone.py:451-461)
    for i in range(stem_down):
        x = conv_norm_relu(x, sd, '%s.patch_embed.convs.%d' % (bb, i), octree, depth - i, '333', 1)
        x = conv_norm_relu(x, sd, '%s.patch_embed.downsamples.%d' % (bb, i), octree, depth - i, '222', 2)
    depth = depth - stem_down
    x = conv_norm_relu(x, sd, bb + '.patch_embed.proj', octree, depth, '333', 1)
    cap['patch_embed'] = x

    plan = WindowPlan(octree, K, dil, max_depth=depth, start_depth=depth - (nlev + noctf) + 1,
                      num_pyramid_levels=nlev, num_octf_levels=noctf, adape_mode=adape)
    cap['plan'] = plan

    # OctFormer stage(s) (hotformerloc_backbone.py:715-718)
    for s in range(noctf):
        for i in range(params.num_blocks[s]):
            x = octformer_block(x, sd, '%s.octf_stage.%d.blocks.%d' % (bb, s, i), plan, depth,
                                heads[s], 1 if i % 2 == 0 else dil)
            cap['octf.%d.%d' % (s, i)] = x
        x = downsample(x, sd, '%s.downsample.%d' % (bb, s), octree, depth)
        depth -= 1
    cap['octf_out'] = x

    # pyramid init (hotformerloc_backbone.py:540-572)
    hs = bb + '.hotf_stage'
    H = heads[noctf] if len(heads) > noctf else heads[-1]
    depths = [depth - j for j in range(nlev)]
    feats, rts = {depths[0]: x}, {}
    for j, d in enumerate(depths):
        rts[d] = relay_token_init(feats[d], sd, hs + '.relay_tokeniser', plan, d,
                                  use_cpe=(adape is None))
        if adape is not None:
            rts[d] = rts[d] + _mlp(plan.window_stats[d], sd, hs + '.rt_adape.mlp')
        if j < nlev - 1:
            feats[d - 1] = downsample(feats[d], sd, '%s.downsamples.%d' % (hs, j), octree, d)
    for d in depths:
        cap['rt_init.%d' % d] = rts[d]
        cap['feat_init.%d' % d] = feats[d]

    # 10 x [RTSA ; 3 x H-OSA] (hotformerloc_backbone.py:593-633)
    for i in range(params.num_blocks[-1]):
        rts = rtsa_block(rts, sd, '%s.rtsa_blocks.%d' % (hs, i), plan, H)
        for j, d in enumerate(depths):
            feats[d], rts[d] = hosa_block(feats[d], rts[d], sd,
                                          '%s.hosa_blocks.%d.%d' % (hs, j, i), plan, d, H)
        if i == 0:
            for d in depths:
                cap['feat_iter0.%d' % d] = feats[d]
                cap['rt_iter0.%d' % d] = rts[d]
    for d in depths:
        cap['feat_final.%d' % d] = feats[d]
        cap['rt_final.%d' % d] = rts[d]

    y = pyramid_attn_pool_mixer(feats, sd, 'pooling.pooling', plan, params.k_pooled_tokens)
    cap['pooled'] = y
    if params.normalize_embeddings:
        y = F.normalize(y, dim=1)
    return y


@torch.no_grad()
def forward(sd: Dict[str, torch.Tensor], params, octree, capture: Optional[dict] = None):
    """Inference forward (no autograd graph); see :func:`forward_with_grad` for the body."""
    return forward_with_grad(sd, params, octree, capture)
