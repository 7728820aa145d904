"""HOTFormerLoc on MI355X: module tree with the reference's parameter names, forward
re-designed around the HIP kernels.

Drop-in contract: `model_factory(ModelParams)` returns a
`torch.nn.Module` whose `forward({'octree': O}) -> {'global': (B, output_dim)}` and whose
`state_dict()` keys/shapes equal the reference's (Appendix D), so reference checkpoints load
with `load_state_dict`.  Reference files followed: `models/hotformerloc.py`,
`models/hotformerloc_backbone.py`, `models/octformer_backbone.py`,
`models/layers/{octformer_layers,pooling,pooling_wrapper,salsa}.py`.

What is different from the reference (same function, different dataflow):
  * tokens are never padded, permuted into windows or concatenated with relay tokens in
    HBM: per-token ops (LayerNorm, Linear, MLP -- hipBLASLt fp32 through PyTorch-ROCm) run
    on the octree-ordered (N_t, C) stream, windows exist only as index arithmetic inside
    the attention kernel (`ops.window_attention`);
  * in the pyramid stage every depth keeps ONE buffer [tokens | relay tokens] so the
    token and relay-token rows share each GEMM launch;
  * relay-token self-attention is ragged per cloud (`ops.relay_attention`), no
    concat/pad/split, no host sync;
  * CPE = depth-wise octree conv + LayerNorm + residual is one kernel (`ops.cpe_forward`).
Only the options the shipped configs use are implemented; anything else raises.
"""

import contextlib
import os
import weakref
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from . import autograd as ag
from . import dwconv as hdw
from . import ops
from .plan import WindowPlan


_FORCE_TRAIN_PATH = False

# Configuration surface.  PRODUCT switches (read from the environment at import, each selects a tested mode):
#   HFL_GEMM = x3 | x6 | fp32 | bf16x3     Linear arithmetic (set_gemm_mode): split precision (default), matched precision, ...
#   HFL_CHECKPOINT = auto | always | never activation checkpointing policy of the training path (set_checkpoint_policy)
#   HFL_CHECKPOINT_FREE_FRACTION           its memory threshold
# Everything else below is a PROBE knob of the A/B scripts under tools/ (profiles/*_ab_*.log): the schedule and fusion choices
# those measurements settled.  They keep their measured defaults unless HFL_PROBES=1 is set in the environment.
_PROBES = os.environ.get('HFL_PROBES', '0') == '1'


def _knob(name: str, default: str) -> str:
    return os.environ.get(name, default) if _PROBES else default



def _grad_path(x=None) -> bool:
    """True when autograd must see the op (training / fine-tuning): route through the
    autograd Functions instead of the fused inference kernels."""
    return torch.is_grad_enabled() or _FORCE_TRAIN_PATH


@contextlib.contextmanager
def training_numerics():
    """Run the training-path kernels even with autograd off.  Stage 1 of the multi-staged step
    (`training/trainer.py:305-317`) must produce the SAME descriptors as stage 3 recomputes with autograd --
    the loss gradient is taken at the stage-1 values, and tau1 = 0.01 amplifies any difference 100x."""
    global _FORCE_TRAIN_PATH
    old, _FORCE_TRAIN_PATH = _FORCE_TRAIN_PATH, True
    try:
        yield
    finally:
        _FORCE_TRAIN_PATH = old


def _ln(x, m: nn.LayerNorm):
    """LayerNorm over channels: HIP kernel in inference, torch (autograd) when grads are needed."""
    if x.is_cuda and not _grad_path(x) and x.shape[-1] in ops._LN_CHANNELS:
        return ops.layer_norm(x, m.weight, m.bias, m.eps)
    if x.is_cuda and _TRAIN_LN and x.numel() > 0 and x.shape[-1] in ops._LN_CHANNELS and x.dtype == torch.float32:
        return ag.layer_norm(x, m.weight, m.bias, m.eps)            # HIP forward + backward
    return F.layer_norm(x, m.normalized_shape, m.weight, m.bias, m.eps)


def _add_ln(x, y, m: nn.LayerNorm):
    """(x + y, LN(x + y)): one fused pass in inference."""
    if x.is_cuda and not _grad_path(x) and x.shape[-1] in ops._LN_CHANNELS:
        return ops.add_layer_norm(x, y, m.weight, m.bias, m.eps)
    x = x + y
    return x, _ln(x, m)


# ---- Linear layers: 'fp32' = hipBLASLt fp32 GEMM; 'bf16x3' = one bf16 GEMM over K-concatenated
# (hi|hi|lo) x (hi|lo|hi) operands with fp32 accumulation/output (include/hotformerloc_hip.h section 9).
# 'x3' = the same split arithmetic on the hand-written kernel (csrc/gemm_x3.hip: pre-split "split2" operands, bias /
# GELU / residual / re-split fused into the epilogue) for the Linear layers of the transformer blocks.
# 'x6' = MATCHED PRECISION: fp32-grade products on the hand-written kernel csrc/gemm_x6.hip (three bf16 planes per operand,
# six plane products, fp32 accumulation: as accurate as an fp32 GEMM), fp32 LayerNorm / softmax / GELU, window attention
# on the fp32 matrix cores -- the reference's own arithmetic (models/layers/octformer_layers.py:53-59,
# models/octformer_backbone.py:52-93) without a library GEMM in the transformer blocks.
_GEMM_MODE = os.environ.get('HFL_GEMM', 'x3')
_PYRAMID_STREAMS = _knob('HFL_PYRAMID_STREAMS', '1') != '0'
_SIDE_STREAM_MAX_ROWS = int(_knob('HFL_SIDE_STREAM_MAX_ROWS', '32768'))
# training-path Linear layers on split-bf16 GEMMs (autograd.LinearSplitFn).  Off by default: parity-tested, but
# at 191 ms/step (B=32, Wild-Places) still slower than the fp32 hipBLASLt route (145 ms) because the operand
# splits of the backward are torch element-wise passes; needs fused split kernels to pay off.
_TRAIN_SPLIT = _knob('HFL_TRAIN_SPLIT', '0') != '0'
# training-path Linear layers on the hand-written split GEMMs (autograd.LinearX3Fn: forward + dx on hfl_linear_x3, dW / db
# on hfl_wgrad_x3)
_TRAIN_X3 = _knob('HFL_TRAIN_X3', '1') != '0'
_ATTN_F16 = _knob('HFL_ATTN_F16', '1') != '0'   # fp16 (hi, lo) MFMA window attention where eligible (A/B switch)
_TRAIN_LN = _knob('HFL_TRAIN_LN', '1') != '0'    # training-path LayerNorm: HIP forward + backward kernels
_TRAIN_MLP = _knob('HFL_TRAIN_MLP', '1') != '0'          # fused fc1 -> GELU -> fc2 autograd Function
_GROUPED_TAPS = _knob('HFL_GROUPED_TAPS', '1') != '0'     # live-tap convolutions: one grouped x3 launch for all taps
_SPARSE_CONV = _knob('HFL_SPARSE_CONV', '1') != '0'    # large 3x3x3 convs over live taps only
_GATHER_IN_GEMM = _knob('HFL_GATHER_IN_GEMM', '1') != '0'   # grouped tap GEMM gathers its A rows itself
_LT_EPILOGUE = _knob('HFL_LT_EPILOGUE', '1') != '0'      # proj / fc2: bias + residual in the GEMM launch
_EARLY_PHASE = _knob('HFL_EARLY_PHASE', '1') != '0'      # token-row half of a block issued before / beside RTSA
_DROP_POOL = _knob('HFL_DROP_POOL', '1') != '0'          # stochastic-depth draws of a forward in one batch of launches
# LN1 -> qkv -> window attention of a relay-token block's token rows as one launch (csrc/attn_ws.hip) from this many token rows
_RTSA_SEGMENTS = _knob('HFL_RTSA_SEGMENTS', '1') != '0'  # RTSA reads the levels' relay rows in place (no torch.cat)
_TRAIN_CPE_FUSED = _knob('HFL_TRAIN_CPE_FUSED', '1') != '0'  # training CPE forward as the fused launch (autograd.CpeFn)
_CPE_FIRST = _knob('HFL_CPE_FIRST', '0') != '0'       # plain schedule: the finest level's CPE before the relay-token block (same stream)
_TRAIN_CPE_BUFFER = _knob('HFL_TRAIN_CPE_BUFFER', '1') != '0'     # probe: 0 = slices + torch.cat around the CPE of a block
_RELAY_IN_PLACE = _knob('HFL_RELAY_IN_PLACE', '1') != '0'  # blocks read RTSA's relay rows in place (no copy launch)
_ATTN_WS = _knob('HFL_ATTN_WS', '1') != '0'
_ATTN_WS_MIN_ROWS = int(_knob('HFL_ATTN_WS_MIN_ROWS', '40000'))
_ATTN_WS_EARLY = _knob('HFL_ATTN_WS_EARLY', '0') != '0'   # keep the early-phase schedule beside it (A/B, tests)
_MERGED_ATTN = _knob('HFL_MERGED_ATTN', '1') != '0'      # window attention of an iteration's levels as one launch
_MAIN_HI = _knob('HFL_MAIN_HI', '0') != '0'             # probe: the inference forward on a high-priority stream


def set_train_split(enabled: bool):
    """Route the training-path Linear layers through the split-bf16 GEMMs (experimental, see _TRAIN_SPLIT)."""
    global _TRAIN_SPLIT
    _TRAIN_SPLIT = bool(enabled)


def set_train_x3(enabled: bool):
    """Training-path Linear layers through `autograd.LinearX3Fn` (default on in GEMM mode 'x3')."""
    global _TRAIN_X3
    _TRAIN_X3 = bool(enabled)


def set_attention_f16(enabled: bool):
    """Window attention on the fp16 (hi, lo) MFMA kernel (qkv written as its operands by the projection GEMM)."""
    global _ATTN_F16
    _ATTN_F16 = bool(enabled)


_SERIAL_STREAMS = False


def set_pyramid_streams(enabled):
    """Run the pyramid depths of one H-OSA iteration on separate HIP streams (inference path).  'serial': the launches of
    that schedule, in its order, all on the current stream (bench.py's roofline leg: a kernel's events then bracket the
    kernel alone)."""
    global _PYRAMID_STREAMS, _SERIAL_STREAMS
    _SERIAL_STREAMS = enabled == 'serial'
    _PYRAMID_STREAMS = bool(enabled)

_W3_CACHE = {}          # id(weight) -> (weakref to weight, version, W3)


def set_gemm_mode(mode: str):
    global _GEMM_MODE
    assert mode in ('fp32', 'bf16x3', 'x3', 'x6')
    _GEMM_MODE = mode


def get_gemm_mode() -> str:
    return _GEMM_MODE


def _split_path(x) -> bool:
    # (channel widths that are not a multiple of the GEMM tile, e.g. 192 in per-level-width configs: fp32 library path)
    return _GEMM_MODE in ('bf16x3', 'x3') and x.is_cuda and not _grad_path() and x.shape[-1] % 128 == 0


def _w2(lin: nn.Linear):
    """split2 layout of a Linear weight for `ops.linear_x3`, cached like `_w3`."""
    w = lin.weight
    key = ('x3', id(w))
    hit = _W3_CACHE.get(key)
    if hit is None or hit[0]() is not w or hit[1] != w._version or hit[3] != w.data_ptr():
        if hit is None or hit[0]() is not w:
            # the split copy is as large as the weight: release it with the parameter (a model that ran inference must
            # not leak its Linear bytes for the life of the process)
            weakref.finalize(w, _W3_CACHE.pop, key, None)
        hit = (weakref.ref(w), w._version, ops.split2_weight(w), w.data_ptr())
        _W3_CACHE[key] = hit
    return hit[2]


def _w6(lin: nn.Linear):
    """The three bf16 planes of a Linear weight for `ops.linear_x6`, cached like `_w2`."""
    w = lin.weight
    key = ('x6', id(w))
    hit = _W3_CACHE.get(key)
    if hit is None or hit[0]() is not w or hit[1] != w._version or hit[3] != w.data_ptr():
        if hit is None or hit[0]() is not w:
            weakref.finalize(w, _W3_CACHE.pop, key, None)
        hit = (weakref.ref(w), w._version, ops.x6_pack(w), w.data_ptr())
        _W3_CACHE[key] = hit
    return hit[2]


def _x6_lin_ok(lin: nn.Linear) -> bool:
    return ops.linear_x6_ok(lin.in_features, lin.out_features)


def _x6_path(x, *linears) -> bool:
    """Matched-precision inference path: GEMM mode 'x6', no autograd, fp32 rows on the GPU, every Linear a shape the kernel
    takes (in_features % 32 == 0, out_features % 128 == 0)."""
    return (_GEMM_MODE == 'x6' and x.is_cuda and not _grad_path() and x.dtype == torch.float32 and x.numel() > 0
            and all(_x6_lin_ok(l) for l in linears))


def _block_tail_x6(x, attn_out, proj: nn.Linear, norm2: nn.LayerNorm, mlp: 'MLP'):
    """x + proj(attn_out) -> x + fc2(gelu(fc1(LN2(x)))) with fp32-grade products: bias + residual ride in the proj / fc2
    launches, bias + GELU in fc1's (four launches: proj, LayerNorm, fc1, fc2)."""
    x = ops.linear_x6(attn_out, _w6(proj), bias=proj.bias, residual=x)
    h = ops.layer_norm(x, norm2.weight, norm2.bias, norm2.eps)
    g = ops.linear_x6(h, _w6(mlp.fc1), bias=mlp.fc1.bias, gelu=True)
    return ops.linear_x6(g, _w6(mlp.fc2), bias=mlp.fc2.bias, residual=x)


# LN2 -> fc1 -> GELU -> fc2 -> residual as ONE launch (csrc/mlp_fused.hip) from this many rows on.  (Rounds 3-4: 24 576 -- below
# that the three-launch form is as fast ALONE.  In the step the coarse pyramid levels' chains are launch-bound: with them on the
# fused launches too -- hidden / feature dimension split over the chip -- three alternating runs gave 2952-2975 clouds/s against
# 2914-2947, profiles/r05_g_ab.log.)
_MLP_FUSED = _knob('HFL_MLP_FUSED', '1') != '0'
_MLP_FUSED_MIN_ROWS = int(_knob('HFL_MLP_FUSED_MIN_ROWS', '1000'))


def _mlp_pack(mlp: 'MLP', rows: int):
    """Weight image of the fused MLP launch for this block, or None when the launch does not apply (channel width, row count,
    missing biases).  Cached per (fc1, fc2) parameter pair like `_w2`."""
    f1, f2 = mlp.fc1, mlp.fc2
    c = f1.in_features
    if not (_MLP_FUSED and rows >= _MLP_FUSED_MIN_ROWS and c in (128, 256) and f1.out_features == 4 * c
            and f2.in_features == 4 * c and f2.out_features == c and f1.bias is not None and f2.bias is not None):
        return None
    w1, w2 = f1.weight, f2.weight
    key = ('mlp', id(w1), id(w2))
    hit = _W3_CACHE.get(key)
    stamp = (w1._version, w2._version, w1.data_ptr(), w2.data_ptr())
    if hit is None or hit[0]() is not w1 or hit[1]() is not w2 or hit[2] != stamp:
        if hit is None or hit[0]() is not w1:
            weakref.finalize(w1, _W3_CACHE.pop, key, None)
        hit = (weakref.ref(w1), weakref.ref(w2), stamp, ops.mlp_fused_pack(w1, w2))
        _W3_CACHE[key] = hit
    return hit[3]


# LN1 -> qkv of the token rows as ONE launch (csrc/qkv_fused.hip) from 24 576 rows on (C = 128, 118 096 rows: 71 vs 96 us for
# LayerNorm + qkv GEMM; C = 256, 65 536 rows: 95 vs 114-122 us).  Round 3 skipped C = 256 shapes whose last round of 32 768 rows
# was less than half full (66 775 rows, 2.04 rounds: three passes, 153 vs 127 us); since the left-over rows are computed with
# the output features split over the workgroups (round 4) that shape takes 101 us and the restriction is gone
# (HFL_QKV_FUSED_MIN_FILL restores it).
_QKV_FUSED = _knob('HFL_QKV_FUSED', '1') != '0'
_QKV_FUSED_MIN_ROWS = int(_knob('HFL_QKV_FUSED_MIN_ROWS', '1000'))
_RTSA_MLP_FUSED = _knob('HFL_RTSA_MLP_FUSED', '1') != '0'
# relay-token block: LN1 -> qkv as ONE launch (csrc/qkv_fused.hip, output features split over the chip for the ~2 k rows) and
# the ragged attention reading its fp16 (hi, lo) rows and writing attention.proj's split2 operand itself
# (hfl_relay_attention_f16_fwd): LayerNorm, qkv GEMM, memset, attention, split2 -> two launches.  The block is a chain of tiny
# launches on the cycle every H-OSA iteration waits for.
_RTSA_SLIM = _knob('HFL_RTSA_SLIM', '1') != '0'
# LN1 -> qkv -> window attention of the blocks without relay tokens (OctFormer stage) as one launch (csrc/attn_fused.hip)
_ATTN_FUSED = _knob('HFL_ATTN_FUSED', '1') != '0'
# attentional pooling of the head as one launch per level (csrc/attn_pool.hip) instead of GEMM + segment softmax + two
# padding copies + batched GEMM
_ATTN_POOL = _knob('HFL_ATTN_POOL', '1') != '0'
# join every pyramid stream at the end of every H-OSA iteration (the schedule of rounds 2-3); 0: only the true dependencies
_ITER_JOIN = _knob('HFL_ITER_JOIN', '0') != '0'
_PLAN_LATE = _knob('HFL_PLAN_LATE', '1') != '0'           # window plan built after the stem has been issued
# relay-token self-attention on a stream of its own (1) or on the finest level's, behind that level's CPE / LN1 / qkv (0)
_RTSA_STREAM = _knob('HFL_RTSA_STREAM', '1') != '0'
_QKV_FUSED_MIN_FILL = float(_knob('HFL_QKV_FUSED_MIN_FILL', '0.0'))


def _qkv_pack(att: 'OctreeAttention', rows: int):
    """Weight image of the fused LN1 -> qkv launch for this block, or None when the launch does not apply / does not pay."""
    lin = att.qkv
    c = lin.in_features
    if not (_QKV_FUSED and rows >= _QKV_FUSED_MIN_ROWS and c in (128, 256) and lin.out_features == 3 * c and lin.bias is not None):
        return None
    if c == 256:
        fill = (rows % 32768) / 32768.0
        if 0.0 < fill < _QKV_FUSED_MIN_FILL:
            return None
    w = lin.weight
    key = ('qkvpack', id(w))
    hit = _W3_CACHE.get(key)
    stamp = (w._version, w.data_ptr())
    if hit is None or hit[0]() is not w or hit[1] != stamp:
        if hit is None or hit[0]() is not w:
            weakref.finalize(w, _W3_CACHE.pop, key, None)
        hit = (weakref.ref(w), stamp, ops.qkv_fused_pack(w))
        _W3_CACHE[key] = hit
    return hit[2]


def _block_tail_x3(x, attn_out2, attn: 'OctreeAttention', norm2: nn.LayerNorm, mlp: 'MLP', fused_any_rows: bool = False):
    """proj (+bias +residual) -> LN2 -> fc1 (+bias, GELU, re-split) -> fc2 (+bias +residual): the proj launch of the
    hand-written GEMM, then the MLP branch as one fused launch (hidden activation in registers) or, for small row counts,
    as LayerNorm + two GEMM launches (the M x 4C hidden activation crosses HBM once each way as 4 B per element)."""
    x = ops.linear_x3(attn_out2, _w2(attn.proj), bias=attn.proj.bias, residual=x)
    pack = _mlp_pack(mlp, _MLP_FUSED_MIN_ROWS if fused_any_rows else x.shape[0])
    if pack is not None:
        return ops.ln_mlp_fused(x, norm2.weight, norm2.bias, norm2.eps, pack, mlp.fc1.bias, mlp.fc2.bias)
    h2 = ops.layer_norm_split2(x, norm2.weight, norm2.bias, norm2.eps)
    g2 = ops.linear_x3(h2, _w2(mlp.fc1), bias=mlp.fc1.bias, gelu_split_out=True)
    return ops.linear_x3(g2, _w2(mlp.fc2), bias=mlp.fc2.bias, residual=x)


def _w3(lin: nn.Linear):
    w = lin.weight
    hit = _W3_CACHE.get(id(w))
    # data_ptr: `module.to(other_device)` swaps `.data` without bumping the version counter
    if hit is None or hit[0]() is not w or hit[1] != w._version or hit[3] != w.data_ptr():
        if len(_W3_CACHE) > 4096:                     # drop entries whose weight is gone
            for k in [k for k, v in _W3_CACHE.items() if v[0]() is None]:
                del _W3_CACHE[k]
        if hit is None or hit[0]() is not w:
            weakref.finalize(w, _W3_CACHE.pop, id(w), None)
        hit = (weakref.ref(w), w._version, ops.split_weight(w), w.data_ptr())
        _W3_CACHE[id(w)] = hit
    return hit[2]


class SplitLinear(nn.Linear):
    """`nn.Linear` whose training-path products use the split-bf16 GEMMs (autograd.LinearSplitFn) when the
    GEMM mode is 'bf16x3'; parameters, names and the fp32 fallback are those of nn.Linear."""

    def forward(self, x):
        if (_GEMM_MODE == 'x3' and _TRAIN_X3 and x.is_cuda and _grad_path() and x.numel() > 0
                and ag.linear_x3_ok(self.in_features, self.out_features)):
            return ag.linear_x3(x, self.weight, self.bias)       # hand-written split GEMM, forward and dx
        if (_GEMM_MODE == 'bf16x3' and _TRAIN_SPLIT and x.is_cuda and _grad_path()
                and self.in_features % 8 == 0 and self.out_features % 8 == 0 and x.numel() > 0):
            return ag.linear_split(x, self.weight, self.bias)
        if _x6_path(x, self):
            y = ops.linear_x6(x.reshape(-1, self.in_features), _w6(self), bias=self.bias)
            return y.view(*x.shape[:-1], self.out_features)
        return F.linear(x, self.weight, self.bias)


_TAP_STREAMS = int(_knob('HFL_TAP_STREAMS', '1'))       # >1: per-tap GEMMs on a stream pool (measured neutral)
_TAP_POOLS = {}


def _tap_stream_pool(device):
    key = (device.type, device.index)
    if key not in _TAP_POOLS:
        _TAP_POOLS[key] = [torch.cuda.Stream(device=device) for _ in range(_TAP_STREAMS)]
    return _TAP_POOLS[key]


def _block_tail_split(x, attn_out3, attn: 'OctreeAttention', norm2: nn.LayerNorm, mlp: 'MLP'):
    """proj -> +residual -> LN2 -> fc1 -> GELU -> fc2 -> +residual with every bias folded into
    the element-wise kernel that follows its GEMM."""
    if _GEMM_MODE == 'x3':
        return _block_tail_x3(x, attn_out3, attn, norm2, mlp)
    if _LT_EPILOGUE:
        # bias + residual ride in the GEMM launch (hfl_gemm_bf16): no pass over the residual stream
        x = ops.gemm_bf16(attn_out3, _w3(attn.proj), bias=attn.proj.bias, residual=x)
        h3 = ops.layer_norm_split3(x, norm2.weight, norm2.bias, norm2.eps)
        g3 = ops.bias_gelu_split3(ops.split_mm(h3, _w3(mlp.fc1)), mlp.fc1.bias)
        return ops.gemm_bf16(g3, _w3(mlp.fc2), bias=mlp.fc2.bias, residual=x)
    p = ops.split_mm(attn_out3, _w3(attn.proj))
    x, h3 = ops.add_layer_norm_split3(x, p, norm2.weight, norm2.bias, norm2.eps, add_bias=attn.proj.bias)
    g3 = ops.bias_gelu_split3(ops.split_mm(h3, _w3(mlp.fc1)), mlp.fc1.bias)
    return ops.add_bias(x, ops.split_mm(g3, _w3(mlp.fc2)), mlp.fc2.bias)


# Activation checkpointing policy.  `grad_checkpoint = True` in the reference's configs buys memory with a second forward of
# every block (written for 24-80 GB devices).  'always': as the reference; 'never': keep the activations; 'auto' (default):
# keep them when they fit comfortably -- an MI355X has 288 GB and the largest shipped training shape (CS-Wild-Places, 64
# clouds) keeps ~40 GB -- and checkpoint like the reference otherwise: a stage keeps its blocks' activations when their
# estimated size is below _CHECKPOINT_FREE_FRACTION of the memory that is free when the stage starts.
# The results are the same either way (the recomputation repeats the same launches on the same inputs).
_CHECKPOINT_POLICY = os.environ.get('HFL_CHECKPOINT', 'auto')
_CHECKPOINT_FREE_FRACTION = float(os.environ.get('HFL_CHECKPOINT_FREE_FRACTION', '0.5'))
# what one transformer block keeps per (row, channel) for its backward: MLP branch x 4 + LN split 4 + GELU split 16 +
# pre-activation 16, attention branch x 4 + LN split 4 + qkv 12 + attention split 4, CPE input and norm ~16 (measured:
# 40.7 GiB against 38 GiB by this count on the config-3 workload)
_BLOCK_BYTES_PER_ROW_CHANNEL = 80


def set_checkpoint_policy(policy: str):
    """'always' | 'never' | 'auto' (see above); returns the previous policy."""
    global _CHECKPOINT_POLICY
    assert policy in ('always', 'never', 'auto')
    prev, _CHECKPOINT_POLICY = _CHECKPOINT_POLICY, policy
    return prev


def _free_memory(device) -> int:
    free, _ = torch.cuda.mem_get_info(device)
    # (what the caching allocator holds but has not handed out is free for this purpose)
    return free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)


def _use_checkpoint(stage, row_channels: int = 0, n_blocks: int = 1, device=None) -> bool:
    """The reference checkpoints every transformer block `if self.grad_checkpoint and self.training`
    (models/octformer_backbone.py:415-416, models/hotformerloc_backbone.py:596-618); here additionally subject to the policy
    above, decided once per forward and stage (all blocks of a stage go the same way): `row_channels` = sum over the stage's
    levels of rows x channels, `n_blocks` = blocks per level."""
    if not (bool(stage.grad_checkpoint) and stage.training and torch.is_grad_enabled()):
        return False
    if _CHECKPOINT_POLICY == 'always':
        return True
    if _CHECKPOINT_POLICY == 'never':
        return False
    if device is None or device.type != 'cuda' or row_channels <= 0:
        return True
    need = row_channels * n_blocks * _BLOCK_BYTES_PER_ROW_CHANNEL
    return need > _CHECKPOINT_FREE_FRACTION * _free_memory(device)


def _checkpoint_block(blk, *args):
    """`checkpoint(blk, *args, use_reentrant=False)` with the block's stochastic-depth factors captured in the closure: the
    recomputation in the backward must see the draws of ITS forward, also when another forward has re-armed the model in
    between (forward A, forward B, backward A).  The reference's checkpoint gets this from the preserved RNG state
    (octformer_backbone.py:415-416); here the draws are tensors handed out by `arm_drop_paths`."""
    mods = blk.__dict__.get('_drop_path_mods')
    if mods is None:
        mods = blk.__dict__['_drop_path_mods'] = [m for m in blk.modules() if isinstance(m, OctreeDropPath)]
    snap = [(m, m._factors) for m in mods]

    def run(*a):
        # the saved draws are visible during the block call only: the recomputation in the backward must not leave the previous
        # step's factors armed on the modules (a submodule called on its own afterwards draws fresh ones)
        prev = [(m, m._factors, m._calls) for m, _ in snap]
        for m, f in snap:
            m._factors, m._calls = f, 0
        try:
            return blk(*a)
        finally:
            for m, f, c in prev:
                m._factors, m._calls = f, c
    return checkpoint(run, *args, use_reentrant=False)


def disarm_drop_paths(model: nn.Module):
    """End of a model forward: the pooled draws belong to that forward only.  A submodule called on its own afterwards
    (backbone alone, tests) draws fresh factors instead of silently reusing the last forward's."""
    cache = model.__dict__.get('_drop_path_cache')
    if cache is not None:
        for m in cache['mods']:
            m._factors = None


def _require_layernorm(conv_norm: str):
    if conv_norm.lower() != 'layernorm':
        raise NotImplementedError("conv_norm=%r: every shipped config uses 'layernorm'" % conv_norm)


def arm_drop_paths(model: nn.Module, batch_size: int, device, dtype=torch.float32):
    """Draw the per-cloud stochastic-depth factors of every active `OctreeDropPath` of `model` for one forward in three
    launches (rand, floor, div over an (instances x 2, B) matrix) and hand each instance its two rows.  Same distribution as
    the reference's per-call draws (models/layers/octformer_layers.py:213-289: independent Bernoulli(keep) per call and
    cloud); the draws themselves cannot match the reference's RNG stream either way."""
    cache = model.__dict__.get('_drop_path_cache')
    if cache is None:
        cache = model.__dict__['_drop_path_cache'] = {'mods': [m for m in model.modules() if isinstance(m, OctreeDropPath)]}
    mods = [m for m in cache['mods'] if m.drop_prob > 0.0]
    sig = (tuple((m.drop_prob, m.scale_by_keep) for m in mods), str(device), dtype)
    if cache.get('sig') != sig:
        keep = [1.0 - m.drop_prob for m in mods for _ in range(2)]
        div = [(1.0 - m.drop_prob) if (m.scale_by_keep and m.drop_prob < 1.0) else 1.0 for m in mods for _ in range(2)]
        cache['sig'] = sig
        cache['keep'] = torch.tensor(keep, dtype=dtype).reshape(-1, 1).to(device)
        cache['div'] = torch.tensor(div, dtype=dtype).reshape(-1, 1).to(device)
    if not mods:
        return
    f = torch.floor(torch.rand(2 * len(mods), batch_size, dtype=dtype, device=device) + cache['keep']) / cache['div']
    for i, m in enumerate(mods):
        m._factors = f[2 * i:2 * i + 2]
        m._calls = 0


class OctreeDropPath(nn.Module):
    """Per-cloud stochastic depth (models/layers/octformer_layers.py:213-289): in training every
    cloud keeps (scaled by 1/keep) or drops a residual branch; identity in eval.  `bid` gives the
    cloud of every row (relay rows carry their window owner, padding rows the last cloud)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep

    # Every block calls its drop path twice per forward (attention branch, MLP branch).  Inside a model forward the draws of
    # ALL instances are made at once (`arm_drop_paths`: one rand + floor + div for the whole network instead of four tiny
    # launches per call, ~570 launches per CS-Wild-Places training step); `_factors` holds this instance's two rows (2, B).
    # A recomputation under gradient checkpointing calls the same two branches again and so reads the same rows.
    _factors = None
    _calls = 0

    def _draw(self, batch_size: int, dtype, device):
        """(B,) per-cloud factor of one call: 0 or 1 / keep."""
        f = self._factors
        if f is not None and f.shape[1] == batch_size and f.device == device and f.dtype == dtype:
            r = f[self._calls % f.shape[0]]
            self._calls += 1
            return r
        keep = 1.0 - self.drop_prob
        rnd = torch.floor(torch.rand(batch_size, dtype=dtype, device=device) + keep)
        if keep > 0.0 and self.scale_by_keep:
            rnd = rnd / keep
        return rnd

    def forward(self, data, bid, batch_size: int):
        if self.drop_prob <= 0.0 or not self.training:
            return data
        return data * self._draw(batch_size, data.dtype, data.device)[bid].unsqueeze(1)

    def row_scale(self, bid, batch_size: int, like):
        """The same draw as forward() as a per-row factor (rows,) for the fused residual branches; None when inactive."""
        if self.drop_prob <= 0.0 or not self.training:
            return None
        return self._draw(batch_size, like.dtype, like.device)[bid].reshape(-1).contiguous()

    def extra_repr(self):
        return 'drop_prob={:.4f}'.format(self.drop_prob)


# --------------------------------------------------------------------------- convs
class OctreeConv(nn.Module):
    """`ocnn.nn.OctreeConv` (nempty=True): parameter `weights` (kdim, Cin, Cout) [+ `bias`].
    gather through the neighbour / child table (HIP) then one fp32 GEMM."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: List[int] = [3],
                 stride: int = 1, nempty: bool = True, use_bias: bool = False):
        super().__init__()
        if not nempty:
            raise NotImplementedError('nempty=False octree conv is off the HOTFormerLoc path')
        ks = list(kernel_size) * 3 if len(kernel_size) == 1 else list(kernel_size)
        self.kernel = ''.join(str(k) for k in ks)
        self.kdim = ks[0] * ks[1] * ks[2]
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.weights = nn.Parameter(torch.empty(self.kdim, in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if use_bias else None
        nn.init.xavier_uniform_(self.weights)

    def takes_split2(self) -> bool:
        """Inference: does forward() run the grouped split-precision GEMM over the live taps, i.e. can it take the split2
        form of its input (rows, 2 Cin) bf16 instead of fp32 rows?"""
        return (_SPARSE_CONV and (self.kernel, self.stride) in (('333', 1), ('222', 2)) and self.in_channels >= 32
                and _GROUPED_TAPS and _GEMM_MODE == 'x3' and self.in_channels % 32 == 0
                and (self.out_channels % 128 == 0 or self.out_channels == 64) and not _grad_path())

    def forward(self, data: torch.Tensor, octree, depth: int):
        if data.dtype == torch.bfloat16:                 # split2 rows from the previous layer's fused norm + ReLU
            assert self.takes_split2() and data.shape[1] == 2 * self.in_channels
            return self._forward_live_taps(data, octree, depth)
        if (_SPARSE_CONV and (self.kernel, self.stride) in (('333', 1), ('222', 2)) and self.in_channels >= 32
                and data.is_cuda):
            if not _grad_path(data):
                return self._forward_live_taps(data, octree, depth)
            if self.in_channels % 64 == 0 and self.out_channels % 64 == 0:      # tile shape of hfl_tap_wgrad
                out = ag.live_tap_conv(data, self.weights, octree, depth, self.kernel, self.stride)
                return out if self.bias is None else out + self.bias
        neigh = octree.get_neigh(depth, self.kernel, self.stride, nempty=True)
        col = ag.octree_gather(data, neigh) if _grad_path(data) else ops.octree_gather(data, neigh)
        w = self.weights.reshape(self.kdim * self.in_channels, self.out_channels)
        if torch.is_grad_enabled() and (col.requires_grad or self.weights.requires_grad):
            return ag.tall_mm(col, w, self.bias)          # (weight gradient contracted over the rows as a split-K product)
        if self.bias is not None:
            return torch.addmm(self.bias, col, w)
        return torch.mm(col, w)


    def _forward_live_taps(self, data, octree, depth):
        """3x3x3 (stride 1) or 2x2x2 (stride 2) conv over the LIVE taps only.  The dense form gathers (N, 27*Cin) -- 80 % zeros on surface-like
        clouds (5.6 live taps of 27 at depth 5, 4 at depth 6) -- and multiplies all of it.  Here: gather one row
        per live (row, tap) pair in tap-major order, one GEMM per tap on its contiguous slice (W[k] is (Cin, Cout)),
        then every output row sums its own partial products through the slot table (the depth-wise conv kernel
        with unit weights: no atomics, fixed summation order)."""
        src, slot, edges = octree.sparse_taps(depth, self.kernel, self.stride)
        if (_GROUPED_TAPS and _GEMM_MODE == 'x3' and self.in_channels % 32 == 0
                and (self.out_channels % 128 == 0 or self.out_channels == 64) and edges[-1] > 0):
            # ONE launch of the split-precision GEMM over all taps: the pairs are gathered from the split2 form of the input
            # (a split2 row is Cin 4-byte cells, so the same gather kernel moves it), row tiles never straddle a tap and
            # carry the offset of their tap's weight block
            npad = max(self.out_channels, 128)
            d2 = data if data.dtype == torch.bfloat16 else ops.split2(data)
            if _GATHER_IN_GEMM and d2.shape[0] * self.in_channels * 4 < (1 << 32):
                # the GEMM's tile loader fetches the pairs' input rows itself: no (pairs, Cin) matrix in memory
                part = ops.linear_x3_grouped_gather(d2, src, self._tap_weights_split2(npad),
                                                    octree.tap_tiles(depth, self.kernel, self.stride, npad), self.out_channels)
            else:
                gs = ops.octree_gather(d2.view(torch.float32), src).view(torch.bfloat16)
                part = ops.linear_x3_grouped(gs, self._tap_weights_split2(npad),
                                             octree.tap_tiles(depth, self.kernel, self.stride, npad), self.out_channels)
            return self._slot_sum(part, slot)
        if (_GEMM_MODE == 'x6' and _GROUPED_TAPS and not _grad_path() and data.dtype == torch.float32 and edges[-1] > 0
                and self.in_channels % 32 == 0 and (self.out_channels % 128 == 0 or self.out_channels == 64)):
            # matched precision: the same grouped launch on hfl_linear_x6 (fp32-grade products; the tile loader gathers the pairs'
            # input rows from the f32 rows themselves)
            npad = max(self.out_channels, 128)
            part = ops.linear_x6_grouped_gather(data, src, self._tap_weights_x6(npad),
                                                octree.tap_tiles(depth, self.kernel, self.stride, npad), self.out_channels)
            return self._slot_sum(part, slot)
        g = ops.octree_gather(data, src)                                  # (P, Cin)
        part = torch.empty((g.shape[0], self.out_channels), dtype=torch.float32, device=data.device)
        w = self.weights
        live = [k for k in range(self.kdim) if edges[k + 1] > edges[k]]
        if _TAP_STREAMS > 1 and len(live) > 2:
            # the per-tap GEMMs are small (thousands of rows each) and independent: deal them over a few HIP streams so
            # that they overlap instead of running one 10-us launch after another
            main = torch.cuda.current_stream()
            pool = _tap_stream_pool(data.device)
            for st in pool:
                st.wait_stream(main)
            for i, k in enumerate(live):
                with torch.cuda.stream(pool[i % len(pool)]):
                    torch.mm(g[edges[k]:edges[k + 1]], w[k], out=part[edges[k]:edges[k + 1]])
            for st in pool:
                main.wait_stream(st)
        else:
            for k in live:
                torch.mm(g[edges[k]:edges[k + 1]], w[k], out=part[edges[k]:edges[k + 1]])
        return self._slot_sum(part, slot)

    def _slot_sum(self, part, slot):
        """every output row adds the partial products of its own live taps (+ the bias) in one pass (hfl_slot_sum)"""
        if part.shape[1] % 4 == 0 and part.shape[1] <= 1024 and slot.shape[1] <= 27 and slot.dtype == torch.int32:
            return ops.slot_sum(part, slot, self.bias)
        out = ops.dwconv_forward_backward(part, self._unit(part.device), slot)
        return out if self.bias is None else out + self.bias

    def _unit(self, device):
        ones = self.__dict__.get('_unit_taps')
        if ones is None or ones.device != device:
            ones = torch.ones((self.kdim, 1, self.out_channels), dtype=torch.float32, device=device)
            self.__dict__['_unit_taps'] = ones
        return ones

    def _tap_weights_x6(self, npad: int):
        """The three bf16 planes of the per-tap weight blocks W[k]^T (Cout x Cin), each padded to `npad` rows, cached per
        parameter version: (3, kdim * npad, Kp) bf16 (`ops.x6_pack`)."""
        w = self.weights
        hit = self.__dict__.get('_w_taps6')
        if hit is None or hit[0] != w._version or hit[1] != w.data_ptr() or hit[2] != npad:
            wt = w.detach().transpose(1, 2)                                   # (kdim, Cout, Cin)
            if npad > self.out_channels:
                wt = torch.cat([wt, wt.new_zeros(self.kdim, npad - self.out_channels, self.in_channels)], 1)
            hit = (w._version, w.data_ptr(), npad, ops.x6_pack(wt.reshape(self.kdim * npad, self.in_channels).contiguous()))
            self.__dict__['_w_taps6'] = hit
        return hit[3]

    def _tap_weights_split2(self, npad: int):
        """split2 layout of the per-tap weight blocks W[k]^T (Cout x Cin), each padded to `npad` rows, cached per parameter
        version: (kdim * npad, 2 Cin) bf16."""
        w = self.weights
        hit = self.__dict__.get('_w_taps')
        if hit is None or hit[0] != w._version or hit[1] != w.data_ptr() or hit[2] != npad:
            wt = w.detach().transpose(1, 2)                                   # (kdim, Cout, Cin)
            if npad > self.out_channels:
                wt = torch.cat([wt, wt.new_zeros(self.kdim, npad - self.out_channels, self.in_channels)], 1)
            hit = (w._version, w.data_ptr(), npad, ops.split2(wt.reshape(self.kdim * npad, self.in_channels).contiguous()))
            self.__dict__['_w_taps'] = hit
        return hit[3]


class OctreeDWConvParams(nn.Module):
    """Holder of the depth-wise conv parameter `weights` (27, 1, C) (reference
    `dwconv.OctreeDWConv`, libs/dwconv/dwconv/nn.py:49-63); the op itself is fused into CPE."""

    def __init__(self, channels: int):
        super().__init__()
        self.weights = nn.Parameter(torch.empty(27, 1, channels))
        nn.init.xavier_uniform_(self.weights)


class OctreeConvNormRelu(nn.Module):
    """models/layers/octformer_layers.py:80-98"""

    def __init__(self, in_channels, out_channels, kernel_size=[3], stride=1, conv_norm='layernorm'):
        super().__init__()
        _require_layernorm(conv_norm)
        self.conv = OctreeConv(in_channels, out_channels, kernel_size, stride, nempty=True)
        self.norm = nn.LayerNorm(out_channels)

    def forward(self, data, octree, depth, split2_out: bool = False):
        """`split2_out` (inference): return the split2 operand of the next convolution's GEMM instead of fp32 rows."""
        y = self.conv(data, octree, depth)
        if y.is_cuda and not _grad_path(y) and y.shape[-1] in ops._LN_CHANNELS:
            return ops.layer_norm_relu(y, self.norm.weight, self.norm.bias, self.norm.eps, split2=split2_out)
        y = F.relu_(_ln(y, self.norm))
        # (a width the fused norm + ReLU kernel is not instantiated for, e.g. 192 of dim = 384: the split as its own pass)
        return ops.split2(y) if split2_out else y


class PatchEmbed(nn.Module):
    """models/octformer_backbone.py:424-461"""

    def __init__(self, in_channels=3, dim=96, num_down=2, conv_norm='layernorm', downsample_input_embeddings=True):
        super().__init__()
        self.num_stages = num_down
        self.downsample_input_embeddings = downsample_input_embeddings
        if not downsample_input_embeddings:            # num_down 3x3x3 convolutions at the input depth (:449-452)
            self.convs = nn.ModuleList([OctreeConvNormRelu(in_channels if i == 0 else dim, dim, [3], 1, conv_norm)
                                        for i in range(num_down)])
            return
        ch = [int(dim * 2 ** i) for i in range(-num_down, 1)]
        self.convs = nn.ModuleList([OctreeConvNormRelu(in_channels if i == 0 else ch[i], ch[i], [3], 1,
                                                       conv_norm) for i in range(num_down)])
        self.downsamples = nn.ModuleList([OctreeConvNormRelu(ch[i], ch[i + 1], [2], 2, conv_norm)
                                          for i in range(num_down)])
        self.proj = OctreeConvNormRelu(ch[-1], dim, [3], 1, conv_norm)

    def forward(self, data, octree, depth):
        if not self.downsample_input_embeddings:
            for i in range(self.num_stages):
                data = self.convs[i](data, octree, depth)
            return data
        # (inference: a layer whose successor runs the grouped split-precision GEMM hands it the split2 rows directly --
        # norm + ReLU + split in one pass instead of three)
        seq = []
        for i in range(self.num_stages):
            seq += [(self.convs[i], depth - i), (self.downsamples[i], depth - i)]
        seq.append((self.proj, depth - self.num_stages))
        fast = data.is_cuda and not _grad_path(data)
        for n, (m, d) in enumerate(seq):
            nxt = seq[n + 1][0].conv if n + 1 < len(seq) else None
            data = m(data, octree, d, split2_out=bool(fast and nxt is not None and nxt.takes_split2()))
        return data


class Downsample(nn.Module):
    """models/octformer_backbone.py:464-477"""

    def __init__(self, in_channels, out_channels, conv_norm='layernorm'):
        super().__init__()
        _require_layernorm(conv_norm)
        self.conv = OctreeConv(in_channels, out_channels, [2], stride=2, nempty=True, use_bias=True)
        self.norm = nn.LayerNorm(out_channels)

    def forward(self, data, octree, depth):
        return _ln(self.conv(data, octree, depth), self.norm)


# ---------------------------------------------------------------- transformer parts
class MLP(nn.Module):
    """models/layers/octformer_layers.py:38-59"""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = SplitLinear(in_features, hidden_features or in_features)
        self.fc2 = SplitLinear(hidden_features or in_features, out_features or in_features)

    def forward(self, x):
        f1, f2 = self.fc1, self.fc2
        if (_GEMM_MODE == 'x3' and _TRAIN_X3 and _TRAIN_MLP and x.is_cuda and _grad_path() and x.numel() > 0
                and f1.bias is not None and f2.bias is not None
                and ag.linear_x3_ok(f1.in_features, f1.out_features) and ag.linear_x3_ok(f2.in_features, f2.out_features)):
            return ag.mlp_x3(x, f1.weight, f1.bias, f2.weight, f2.bias)      # GELU and its gradient inside the GEMMs
        if _x6_path(x, f1, f2) and x.dim() == 2:
            return ops.linear_x6(ops.linear_x6(x, _w6(f1), bias=f1.bias, gelu=True), _w6(f2), bias=f2.bias)
        return self.fc2(F.gelu(self.fc1(x)))


def _mlp_branch(x, norm: nn.LayerNorm, mlp: 'MLP', row_scale=None):
    """x + mlp(LN(x)) on the training path: one fused autograd Function when the shapes allow (LayerNorm writes the GEMM
    operand, GELU and the residual ride in the GEMM epilogues, the skip gradient joins inside the LayerNorm backward)."""
    f1, f2 = mlp.fc1, mlp.fc2
    if (_GEMM_MODE == 'x3' and _TRAIN_X3 and _TRAIN_MLP and _TRAIN_LN and x.is_cuda and x.numel() > 0
            and x.dtype == torch.float32 and x.shape[-1] in ops._LN_CHANNELS and f1.bias is not None and f2.bias is not None
            and ag.linear_x3_ok(f1.in_features, f1.out_features) and ag.linear_x3_ok(f2.in_features, f2.out_features)):
        return ag.ln_mlp_residual_x3(x, norm.weight, norm.bias, norm.eps, f1.weight, f1.bias, f2.weight, f2.bias, row_scale)
    y = mlp(_ln(x, norm))
    return x + (y if row_scale is None else y * row_scale.unsqueeze(1))


class CPE(nn.Module):
    """models/layers/octformer_layers.py:122-142.  xcpe=False: depth-wise octree conv + LayerNorm (one fused kernel in
    inference); xcpe=True (PointTransformerV3's xCPE): full octree conv with bias + Linear + LayerNorm."""

    def __init__(self, dim, conv_norm='layernorm', xcpe=False):
        super().__init__()
        _require_layernorm(conv_norm)
        self.xcpe = xcpe
        if xcpe:
            self.conv = OctreeConv(dim, dim, [3], 1, nempty=True, use_bias=True)
            self.linear = nn.Linear(dim, dim)
        else:
            self.conv = OctreeDWConvParams(dim)
            self.linear = nn.Identity()
        self.norm = nn.LayerNorm(dim)

    def forward(self, data, plan: WindowPlan, depth: int, residual: bool, out=None):
        if self.xcpe:
            y = _ln(self.linear(self.conv(data, plan.octree, depth)), self.norm)
            y = data + y if residual else y
            if out is not None:
                out.copy_(y)
                return out
            return y
        if (_TRAIN_CPE_FUSED and _grad_path(data) and data.shape[1] in (32, 64, 128, 256) and data.is_cuda
                and data.dtype == torch.float32 and out is None):
            # training: the fused launch as the forward (it also writes the convolution's output for the backward)
            return ag.cpe(data, self.conv.weights, self.norm.weight, self.norm.bias, plan.neigh(depth), residual,
                          self.norm.eps)
        if _grad_path(data) or data.shape[1] not in (32, 64, 128, 256):
            # dwconv (HIP fwd/bwd, libs/dwconv semantics) -> LayerNorm -> residual (also the inference path of channel
            # widths the fused kernel is not instantiated for, e.g. 192 in per-level-width configurations)
            y = _ln(hdw.octree_dwconv(data, self.conv.weights, plan.neigh(depth)), self.norm)
            y = data + y if residual else y
            if out is not None:
                out.copy_(y)
                return out
            return y
        return ops.cpe_forward(data, self.conv.weights, self.norm.weight, self.norm.bias,
                               plan.neigh(depth), residual, self.norm.eps, out=out)


class RPE(nn.Module):
    """models/layers/octformer_layers.py:144-174: only the table; the gather happens in
    registers inside the attention kernel."""

    def __init__(self, patch_size, num_heads, dilation=1):
        super().__init__()
        self.pos_bnd = int(0.8 * patch_size * dilation ** 0.5)
        self.rpe_num = 2 * self.pos_bnd + 1
        self.rpe_table = nn.Parameter(torch.zeros(3 * self.rpe_num, num_heads))
        nn.init.trunc_normal_(self.rpe_table, std=0.02)


class OctreeAttention(nn.Module):
    """models/octformer_backbone.py:24-106"""

    def __init__(self, dim, patch_size, num_heads, dilation=1, rt_per_window=0, use_rpe=True):
        super().__init__()
        if dim // num_heads != 16:
            raise NotImplementedError('head dim must be 16 (all shipped configs)')
        self.dim, self.patch_size, self.num_heads = dim, patch_size, num_heads
        self.dilation, self.rt_per_window = dilation, rt_per_window
        self.qkv = SplitLinear(dim, dim * 3)
        self.proj = SplitLinear(dim, dim)
        self.rpe = RPE(patch_size, num_heads, dilation) if use_rpe else None

    def core(self, qkv, plan: WindowPlan, depth: int, qkv_bias=None, out_split=False, qkv_f16=False):
        nt = plan.n_tokens[depth]
        cfg = dict(n_tokens=nt, n_windows=plan.n_windows[depth], patch_size=self.patch_size,
                   dilation=self.dilation, n_relay=self.rt_per_window, n_heads=self.num_heads,
                   batch_size=plan.B, rt_row0=nt, depth=depth)
        table = None if self.rpe is None else self.rpe.rpe_table
        if _grad_path(qkv):
            return ag.window_attention(qkv, table, plan.meta[depth], **cfg)
        return ops.window_attention(qkv, plan.meta[depth], table, qkv_bias=qkv_bias,
                                    out_split=out_split, qkv_f16=qkv_f16, **cfg)

    def forward(self, x, plan: WindowPlan, depth: int):
        """x: (N_t [+ W], C) token rows [followed by the relay-token rows]."""
        return self.proj(self.core(self.qkv(x), plan, depth))

    def residual_branch(self, x, norm1: nn.LayerNorm, plan: WindowPlan, depth: int, row_scale=None):
        """x + attention(LN(x)) on the training path: one fused autograd Function when the shapes allow."""
        C = self.dim
        if (_GEMM_MODE == 'x3' and _TRAIN_X3 and _TRAIN_MLP and _TRAIN_LN and x.is_cuda and x.numel() > 0
                and x.dtype == torch.float32 and C in ops._LN_CHANNELS and ag.linear_x3_ok(C, 3 * C)
                and ag.linear_x3_ok(C, C) and self.proj.bias is not None):
            nt = plan.n_tokens[depth]
            cfg = dict(n_tokens=nt, n_windows=plan.n_windows[depth], patch_size=self.patch_size,
                       dilation=self.dilation, n_relay=self.rt_per_window, n_heads=self.num_heads,
                       batch_size=plan.B, rt_row0=nt, depth=depth)
            table = None if self.rpe is None else self.rpe.rpe_table
            return ag.ln_attn_residual_x3(x, norm1.weight, norm1.bias, norm1.eps, self.qkv.weight, self.qkv.bias, table,
                                          plan.meta[depth], cfg, self.proj.weight, self.proj.bias, row_scale)
        y = self.forward(_ln(x, norm1), plan, depth)
        return x + (y if row_scale is None else y * row_scale.unsqueeze(1))

    def forward_split(self, x, norm1: nn.LayerNorm, plan: WindowPlan, depth: int):
        """LN1 -> qkv -> attention, split-precision path; returns the bf16 operand of `proj`."""
        if _GEMM_MODE == 'x3':         # qkv bias folded into the GEMM epilogue, attention writes split2 rows
            f16 = _ATTN_F16 and ops.window_attention_f16_ok(x.shape[0], self.patch_size, self.dilation,
                                                            self.rt_per_window, self.num_heads, depth)
            qs = 16 ** -0.5 * 1.4426950408889634
            nt = plan.n_tokens[depth]
            qpack = _qkv_pack(self, nt) if (f16 and x.dtype == torch.float32 and x.is_contiguous()) else None
            if qpack is not None:
                # as hfl_block_forward_x3 does it: LN1 -> qkv of the token rows in one launch, the relay rows through
                # LayerNorm + the qkv GEMM
                qkv = torch.empty((x.shape[0], 3 * x.shape[1]), dtype=torch.float32, device=x.device)
                ops.ln_qkv_fused(x[:nt], norm1.weight, norm1.bias, norm1.eps, qpack, self.qkv.bias, qs, out=qkv[:nt])
                if x.shape[0] > nt:          # the relay rows: the same launch, output features split over the workgroups
                    ops.ln_qkv_fused(x[nt:], norm1.weight, norm1.bias, norm1.eps, qpack, self.qkv.bias, qs, out=qkv[nt:])
                return self.core(qkv, plan, depth, out_split=2, qkv_f16=True)
            a2 = ops.layer_norm_split2(x, norm1.weight, norm1.bias, norm1.eps)
            if f16:
                # the projection writes q, k, v as fp16 (hi, lo) MFMA operands (q pre-scaled): fp16-MFMA window kernel
                qkv = ops.linear_x3_qkv(a2, _w2(self.qkv), self.qkv.bias, qs)
                return self.core(qkv, plan, depth, out_split=2, qkv_f16=True)
            qkv = ops.linear_x3(a2, _w2(self.qkv), bias=self.qkv.bias)
            return self.core(qkv, plan, depth, out_split=2)
        a3 = ops.layer_norm_split3(x, norm1.weight, norm1.bias, norm1.eps)
        return self.core(ops.split_mm(a3, _w3(self.qkv)), plan, depth, qkv_bias=self.qkv.bias,
                         out_split=True)


def _drops(block) -> bool:
    """Stochastic depth is live (train mode, non-zero probability): the fused inference paths do not apply it."""
    return block.training and block.drop_path.drop_prob > 0.0


def _init_layer_scale(block, dim, layer_scale):
    """`gamma1` / `gamma2`: learnable channel-wise multipliers of the attention / MLP branches when `layer_scale` is a
    number (models/octformer_backbone.py:214-229), the constant 1 otherwise (not parameters then, as in the reference)."""
    block.use_layer_scale = layer_scale is not None and type(layer_scale) in (int, float)
    if block.use_layer_scale:
        block.gamma1 = nn.Parameter(layer_scale * torch.ones(dim))
        block.gamma2 = nn.Parameter(layer_scale * torch.ones(dim))
    else:
        block.gamma1 = block.gamma2 = 1


_NATIVE_BLOCK = _knob('HFL_NATIVE_BLOCK', '1') != '0'     # inference blocks as one native call (hfl_block_forward_x3)


_F16_OK_CACHE = {}


def _attn_f16_ok(rows, att, depth) -> bool:
    key = (rows, att.patch_size, att.dilation, att.rt_per_window, att.num_heads, depth)
    hit = _F16_OK_CACHE.get(key)
    if hit is None:
        if len(_F16_OK_CACHE) > 4096:
            _F16_OK_CACHE.clear()
        hit = _F16_OK_CACHE[key] = ops.window_attention_f16_ok(*key)
    return hit


def _native_block_static(block, device):
    """(BlockWeights, tensors it points to) of a block, or None when the block's parameters do not qualify for the native
    call; cached on the block and revalidated by the (version, pointer) stamp of every parameter (a few microseconds instead
    of rebuilding a 20-field ctypes structure per block and forward: the host runs only just ahead of the GPU)."""
    att, mlp, cpe = block.attention, block.mlp, block.cpe
    table = None if att.rpe is None else att.rpe.rpe_table
    plist = (cpe.conv.weights, cpe.norm.weight, cpe.norm.bias, block.norm1.weight, block.norm1.bias, block.norm2.weight,
             block.norm2.bias, att.qkv.bias, att.proj.bias, mlp.fc1.bias, mlp.fc2.bias, att.qkv.weight, att.proj.weight,
             mlp.fc1.weight, mlp.fc2.weight, table)
    stamp = tuple((p._version, p.data_ptr()) if p is not None else None for p in plist)
    hit = block.__dict__.get('_native_static')
    if hit is not None and hit[0] == stamp:
        return hit[1]
    from ._native import BlockWeights
    res = None
    # the native call reads raw pointers: every parameter must be what the Python wrappers would have checked (fp32,
    # contiguous, on this device), the biases must exist and the three LayerNorms must share one eps
    ok = (all(p is not None for p in plist[:-1]) and block.norm1.eps == block.norm2.eps == cpe.norm.eps
          and all(p is None or (p.dtype == torch.float32 and p.is_contiguous() and p.device == device) for p in plist))
    if ok:
        keep = [_w2(att.qkv), _w2(att.proj), None, None]
        w = BlockWeights(channels=att.dim, eps=block.norm1.eps, q_scale=16 ** -0.5 * 1.4426950408889634,
                         cpe_weight=cpe.conv.weights.data_ptr(), cpe_gamma=cpe.norm.weight.data_ptr(),
                         cpe_beta=cpe.norm.bias.data_ptr(), norm1_gamma=block.norm1.weight.data_ptr(),
                         norm1_beta=block.norm1.bias.data_ptr(), norm2_gamma=block.norm2.weight.data_ptr(),
                         norm2_beta=block.norm2.bias.data_ptr(), qkv_w=keep[0].data_ptr(), proj_w=keep[1].data_ptr(),
                         fc1_w=None, fc2_w=None, mlp_pack=None, qkv_b=att.qkv.bias.data_ptr(),
                         proj_b=att.proj.bias.data_ptr(), fc1_b=mlp.fc1.bias.data_ptr(), fc2_b=mlp.fc2.bias.data_ptr(),
                         rpe_table=None if table is None else table.data_ptr())
        res = (w, keep)
    block.__dict__['_native_static'] = (stamp, res)
    return res


def _attn_ws_wanted(att, rows: int, nt: int, n_windows: int, depth: int) -> bool:
    """Whether a relay-token block of this shape takes the one-launch LN1 -> qkv -> window attention (csrc/attn_ws.hip)."""
    return (_ATTN_WS and att.rpe is not None and rows > nt >= _ATTN_WS_MIN_ROWS and att.dilation == 1
            and att.rt_per_window == 1 and ops.attn_ws_ok(nt, n_windows, att.patch_size, att.num_heads, depth, att.dim))


def _native_block_call(block, x_in, plan: WindowPlan, depth: int):
    """A prepared native call for the block's inference forward (ops.BlockCall), or None when this block / launch is not
    eligible (then the Python sequence of the same kernels runs).  Same kernels, same order, same results; only the host
    work differs."""
    att = block.attention
    C = att.dim
    if not (_NATIVE_BLOCK and _GEMM_MODE == 'x3' and _ATTN_F16 and _split_path(x_in) and ops.KernelTimer.active is None
            and not block.use_layer_scale and not _drops(block) and not block.cpe.xcpe and C % 128 == 0
            and x_in.dtype == torch.float32 and x_in.is_contiguous()):
        return None
    nt = plan.n_tokens[depth]
    rows = x_in.shape[0]
    if not _attn_f16_ok(rows, att, depth):
        return None
    static = _native_block_static(block, x_in.device)
    if static is None:
        return None
    w, keep = static
    from ._native import WindowAttnDesc
    table = None if att.rpe is None else att.rpe.rpe_table
    bnd = int(0.8 * att.patch_size * att.dilation ** 0.5)
    expanded = None if table is None else ops.rpe_expand(table, att.num_heads, bnd, depth, True)
    if table is not None and expanded is None:
        return None
    mlp = block.mlp
    qpack = _qkv_pack(att, nt)
    w.qkv_pack = None if qpack is None else qpack.data_ptr()
    w.fuse_attention = (1 if _ATTN_FUSED else 0) | (0 if _RELAY_IN_PLACE else 4)
    tables3 = None
    if qpack is not None and _attn_ws_wanted(att, rows, nt, plan.n_windows[depth], depth):
        tables3 = ops.rpe_expand(table, att.num_heads, bnd, depth, 2)
        w.fuse_attention |= 2
    w.rpe_tables3 = None if tables3 is None else tables3.data_ptr()
    pack = _mlp_pack(mlp, rows)
    if pack is not None:
        w.mlp_pack, w.fc1_w, w.fc2_w = pack.data_ptr(), None, None
    else:
        if keep[2] is None:
            keep[2], keep[3] = _w2(mlp.fc1), _w2(mlp.fc2)
        w.mlp_pack, w.fc1_w, w.fc2_w = None, keep[2].data_ptr(), keep[3].data_ptr()
    desc = WindowAttnDesc(n_tokens=nt, rt_row0=nt, n_windows=plan.n_windows[depth], patch_size=att.patch_size,
                          dilation=att.dilation, n_relay=att.rt_per_window, n_heads=att.num_heads, pos_bnd=bnd,
                          batch_size=plan.B, scale=16 ** -0.5, depth=depth,
                          rpe_expanded=None if expanded is None else expanded.data_ptr())
    return ops.BlockCall(w, (keep, expanded, pack, qpack, tables3), x_in, plan.neigh(depth), plan.meta[depth], nt, desc)


def _native_block(block, x_in, relay, plan: WindowPlan, depth: int):
    """The block's inference forward as ONE native call, or None when not eligible (see _native_block_call)."""
    call = _native_block_call(block, x_in, plan, depth)
    return None if call is None else call.run(0, relay)


class OctFormerBlock(nn.Module):
    """models/octformer_backbone.py:182-299 (use_rt=False)"""

    def __init__(self, dim, num_heads, patch_size, dilation, disable_RPE=False, conv_norm='layernorm',
                 drop_path=0.0, layer_scale=None, xcpe=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attention = OctreeAttention(dim, patch_size, num_heads, dilation, 0, not disable_RPE)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, int(dim * 4.0), dim)
        self.drop_path = OctreeDropPath(drop_path)
        self.cpe = CPE(dim, conv_norm, xcpe)
        _init_layer_scale(self, dim, layer_scale)

    def forward(self, x, plan: WindowPlan, depth: int):
        if not _grad_path(x):
            y = _native_block(self, x, None, plan, depth)
            if y is not None:
                return y
        x = self.cpe(x, plan, depth, residual=True)
        if _split_path(x) and not self.use_layer_scale and not _drops(self):
            o3 = self.attention.forward_split(x, self.norm1, plan, depth)
            return _block_tail_split(x, o3, self.attention, self.norm2, self.mlp)
        if self.use_layer_scale:
            bid = plan.row_cloud(depth, with_relay=False)
            x = x + self.drop_path(self.gamma1 * self.attention(_ln(x, self.norm1), plan, depth), bid, plan.B)
            return x + self.drop_path(self.gamma2 * self.mlp(_ln(x, self.norm2)), bid, plan.B)
        if self.training and self.drop_path.drop_prob > 0.0:        # stochastic depth: per-row factor inside the branches
            bid = plan.row_cloud(depth, with_relay=False)
            x = self.attention.residual_branch(x, self.norm1, plan, depth, self.drop_path.row_scale(bid, plan.B, x))
            return _mlp_branch(x, self.norm2, self.mlp, self.drop_path.row_scale(bid, plan.B, x))
        if _grad_path(x):
            return _mlp_branch(self.attention.residual_branch(x, self.norm1, plan, depth), self.norm2, self.mlp)
        att = self.attention
        if _x6_path(x, att.qkv, att.proj, self.mlp.fc1, self.mlp.fc2) and x.shape[-1] in ops._LN_CHANNELS:
            o = att.core(att.qkv(_ln(x, self.norm1)), plan, depth)
            return _block_tail_x6(x, o, att.proj, self.norm2, self.mlp)
        x, h = _add_ln(x, self.attention(_ln(x, self.norm1), plan, depth), self.norm2)
        return x + self.mlp(h)


class OctFormerStage(nn.Module):
    """models/octformer_backbone.py:363-421"""

    def __init__(self, dim, num_heads, patch_size, dilation, num_blocks, disable_RPE=False,
                 conv_norm='layernorm', drop_path=0.0, grad_checkpoint=False, layer_scale=None, xcpe=False):
        super().__init__()
        dp = drop_path if isinstance(drop_path, (list, tuple)) else [drop_path] * num_blocks
        self.grad_checkpoint = grad_checkpoint
        self.blocks = nn.ModuleList([
            OctFormerBlock(dim, num_heads, patch_size, 1 if i % 2 == 0 else dilation, disable_RPE,
                           conv_norm, dp[i], layer_scale, xcpe) for i in range(num_blocks)])

    def forward(self, x, plan, depth):
        ckpt = _use_checkpoint(self, x.shape[0] * x.shape[1], len(self.blocks), x.device)
        for blk in self.blocks:
            # activation checkpointing per block, non-reentrant, as octformer_backbone.py:415-416
            x = _checkpoint_block(blk, x, plan, depth) if ckpt else blk(x, plan, depth)
        return x


class HOTFormerBlock(nn.Module):
    """models/hotformerloc_backbone.py:130-236 (rt_propagation off): operates on the
    [tokens | relay tokens] buffer of one depth."""

    def __init__(self, dim, num_heads, patch_size, disable_RPE=False, conv_norm='layernorm',
                 drop_path=0.0, layer_scale=None, xcpe=False, rt_propagation=False, rt_propagation_scale=None, last=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attention = OctreeAttention(dim, patch_size, num_heads, 1, 1, not disable_RPE)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, int(dim * 4.0), dim)
        self.drop_path = OctreeDropPath(drop_path)
        self.cpe = CPE(dim, conv_norm, xcpe)
        _init_layer_scale(self, dim, layer_scale)
        # relay-token propagation (hotformerloc_backbone.py:183-193,224-235): the LAST block of a level adds every window's
        # relay token, times a scalar, to the local features of that window
        self.propagate = bool(last and rt_propagation)
        if self.propagate:
            self.upsampler = nn.Upsample(scale_factor=patch_size, mode='nearest')      # parameter-free; kept for the module tree
            self.rt_gamma_propagate = (nn.Parameter(torch.tensor(float(rt_propagation_scale)))
                                       if rt_propagation_scale is not None and type(rt_propagation_scale) in (int, float) else 1)

    def forward(self, buf, plan: WindowPlan, depth: int, relay=None):
        """buf: [tokens | relay rows] of this depth; `relay` (optional): this depth's relay rows as RTSA just produced
        them -- they replace buf's relay rows without a separate copy into buf first."""
        return self._tail(self._forward(buf, plan, depth, relay), plan, depth)

    def _tail(self, out, plan: WindowPlan, depth: int):
        if not self.propagate:
            return out
        # data + gamma * rt[window of the token], zero where the token's cloud is not the window's owner (rt_init_mask,
        # models/octree.py:143-145): a gather over the relay rows, differentiable torch ops (last block of a level only)
        nt = plan.n_tokens[depth]
        win, keep = plan.token_window(depth)
        tok = out[:nt] + self.rt_gamma_propagate * (out[nt:].index_select(0, win) * keep)
        return torch.cat([tok, out[nt:]], 0)

    def _forward(self, buf, plan: WindowPlan, depth: int, relay=None):
        nt = plan.n_tokens[depth]
        if _grad_path(buf):
            c = self.cpe
            if (not c.xcpe and _TRAIN_CPE_FUSED and _TRAIN_CPE_BUFFER and buf.shape[1] in (32, 64, 128, 256) and buf.is_cuda
                    and buf.dtype == torch.float32 and nt > 0):
                # the CPE launch writes the token rows of the new [tokens | relay rows] buffer (no slices, no concatenation)
                buf = ag.cpe_buffer(buf, relay, c.conv.weights, c.norm.weight, c.norm.bias, plan.neigh(depth), nt, c.norm.eps)
            else:
                buf = torch.cat([self.cpe(buf[:nt], plan, depth, residual=True), buf[nt:] if relay is None else relay], 0)
        else:                                   # CPE writes straight into the new buffer's token rows
            y = _native_block(self, buf, relay, plan, depth)
            if y is not None:
                return y
            new = torch.empty_like(buf)
            self.cpe(buf[:nt], plan, depth, residual=True, out=new[:nt])
            new[nt:].copy_(buf[nt:] if relay is None else relay)
            buf = new
        if _split_path(buf) and not self.use_layer_scale and not _drops(self):
            o3 = self.attention.forward_split(buf, self.norm1, plan, depth)
            return _block_tail_split(buf, o3, self.attention, self.norm2, self.mlp)
        if self.use_layer_scale:
            bid = plan.row_cloud(depth, with_relay=True)
            buf = buf + self.drop_path(self.gamma1 * self.attention(_ln(buf, self.norm1), plan, depth), bid, plan.B)
            return buf + self.drop_path(self.gamma2 * self.mlp(_ln(buf, self.norm2)), bid, plan.B)
        if self.training and self.drop_path.drop_prob > 0.0:
            bid = plan.row_cloud(depth, with_relay=True)
            buf = self.attention.residual_branch(buf, self.norm1, plan, depth, self.drop_path.row_scale(bid, plan.B, buf))
            return _mlp_branch(buf, self.norm2, self.mlp, self.drop_path.row_scale(bid, plan.B, buf))
        if _grad_path(buf):
            return _mlp_branch(self.attention.residual_branch(buf, self.norm1, plan, depth), self.norm2, self.mlp)
        att = self.attention
        if _x6_path(buf, att.qkv, att.proj, self.mlp.fc1, self.mlp.fc2) and buf.shape[-1] in ops._LN_CHANNELS:
            o = att.core(att.qkv(_ln(buf, self.norm1)), plan, depth)
            return _block_tail_x6(buf, o, att.proj, self.norm2, self.mlp)
        buf, h = _add_ln(buf, self.attention(_ln(buf, self.norm1), plan, depth), self.norm2)
        return buf + self.mlp(h)


class RTAttention(nn.Module):
    """models/hotformerloc_backbone.py:56-127"""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.qkv = SplitLinear(dim, dim * 3)
        self.proj = SplitLinear(dim, dim)

    def forward(self, rt, plan: WindowPlan):
        qkv = self.qkv(rt)
        if _grad_path(qkv):
            if plan.max_seq_len <= 611:             # LDS capacity of the HIP backward
                return self.proj(ag.relay_attention(qkv, plan, self.num_heads))
            return self.proj(ag.relay_attention_torch(qkv, plan, self.num_heads))
        out = ops.relay_attention(qkv, plan.seq_rows, plan.seq_off, plan.B, self.num_heads,
                                  plan.max_seq_len)
        return self.proj(out)


class RelayTokenTransformerBlock(nn.Module):
    """models/hotformerloc_backbone.py:239-302 on the concatenated (sum W_d, C) relay rows."""

    def __init__(self, dim, num_heads, drop_path=0.0, layer_scale=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.rt_attention = RTAttention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLP(dim, int(dim * 4.0), dim)
        self.drop_path = OctreeDropPath(drop_path)
        _init_layer_scale(self, dim, layer_scale)                       # hotformerloc_backbone.py:260-272

    def _native_static(self, device):
        """(RelayBlockWeights, tensors it points to) or None, cached and revalidated like _native_block_static."""
        att, mlp = self.rt_attention, self.mlp
        plist = (self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, att.qkv.bias, att.proj.bias,
                 mlp.fc1.bias, mlp.fc2.bias, att.qkv.weight, att.proj.weight, mlp.fc1.weight, mlp.fc2.weight)
        stamp = tuple((p._version, p.data_ptr()) if p is not None else None for p in plist)
        hit = self.__dict__.get('_native_cache')
        if hit is not None and hit[0] == stamp:
            return hit[1]
        res = None
        if (all(p is not None and p.dtype == torch.float32 and p.is_contiguous() and p.device == device for p in plist)
                and self.norm1.eps == self.norm2.eps and att.dim % 128 == 0 and att.num_heads * 16 == att.dim):
            from ._native import RelayBlockWeights
            # (the MLP branch as the fused launch with the hidden dimension split over the chip: 27 us against 48 us for the
            # three launches alone, and inside the step the relay tokens' fc2 -- K = 1024 over 28 workgroups -- took 120 us:
            # 2361 -> 2417 clouds/s, three alternating runs each)
            mpack = _mlp_pack(mlp, _MLP_FUSED_MIN_ROWS) if (_RTSA_MLP_FUSED and att.dim in (128, 256)) else None
            # (LN1 -> qkv as one launch and the attention writing proj's operand: six launches -> three, see _RTSA_SLIM)
            qpack = _qkv_pack(att, _QKV_FUSED_MIN_ROWS) if _RTSA_SLIM else None
            keep = (_w2(att.qkv), _w2(att.proj), _w2(mlp.fc1), _w2(mlp.fc2), mpack, qpack)
            w = RelayBlockWeights(channels=att.dim, n_heads=att.num_heads, eps=self.norm1.eps,
                                  mlp_pack=None if mpack is None else mpack.data_ptr(),
                                  qkv_pack=None if qpack is None else qpack.data_ptr(),
                                  norm1_gamma=self.norm1.weight.data_ptr(), norm1_beta=self.norm1.bias.data_ptr(),
                                  norm2_gamma=self.norm2.weight.data_ptr(), norm2_beta=self.norm2.bias.data_ptr(),
                                  qkv_w=keep[0].data_ptr(), proj_w=keep[1].data_ptr(), fc1_w=keep[2].data_ptr(),
                                  fc2_w=keep[3].data_ptr(), qkv_b=att.qkv.bias.data_ptr(), proj_b=att.proj.bias.data_ptr(),
                                  fc1_b=mlp.fc1.bias.data_ptr(), fc2_b=mlp.fc2.bias.data_ptr())
            res = (w, keep)
        self.__dict__['_native_cache'] = (stamp, res)
        return res

    def forward_parts(self, parts, plan):
        """forward(torch.cat(parts)) -- without the concatenation launch when the native call can read the rows where they are
        (hfl_relay_block_io.x_segments: the fused LN1 -> qkv launch and proj's residual take a row-segment table)."""
        p0 = parts[0]
        if (_RTSA_SEGMENTS and len(parts) <= 4 and _GEMM_MODE == 'x3' and _split_path(p0) and not self.use_layer_scale
                and not _drops(self) and _NATIVE_BLOCK and ops.KernelTimer.active is None
                and all(p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] > 0 for p in parts)):
            static = self._native_static(p0.device)
            if static is not None and static[0].qkv_pack:
                return ops.relay_block_forward_x3(static[0], static[1], list(parts), plan.seq_rows, plan.seq_off, plan.B,
                                                  plan.max_seq_len, plan.orphan_rows)
        return self(torch.cat(list(parts), 0), plan)

    def forward(self, rt, plan):
        if _GEMM_MODE == 'x3' and _split_path(rt) and not self.use_layer_scale and not _drops(self) and rt.shape[0] > 0:
            if _NATIVE_BLOCK and ops.KernelTimer.active is None and rt.dtype == torch.float32 and rt.is_contiguous():
                static = self._native_static(rt.device)          # the same nine launches from ONE native call
                if static is not None:
                    return ops.relay_block_forward_x3(static[0], static[1], rt, plan.seq_rows, plan.seq_off, plan.B,
                                                      plan.max_seq_len, plan.orphan_rows)
            att = self.rt_attention
            qpack = _qkv_pack(att, _QKV_FUSED_MIN_ROWS) if _RTSA_SLIM else None
            if qpack is not None:
                qkv = ops.ln_qkv_fused(rt, self.norm1.weight, self.norm1.bias, self.norm1.eps, qpack, att.qkv.bias,
                                       0.25 * 1.4426950408889634)
                o2 = ops.relay_attention_f16(qkv, plan.seq_rows, plan.seq_off, plan.B, att.num_heads, plan.max_seq_len,
                                             plan.orphan_rows)
            else:
                a2 = ops.layer_norm_split2(rt, self.norm1.weight, self.norm1.bias, self.norm1.eps)
                qkv = ops.linear_x3(a2, _w2(att.qkv), bias=att.qkv.bias)
                o2 = ops.split2(ops.relay_attention(qkv, plan.seq_rows, plan.seq_off, plan.B, att.num_heads, plan.max_seq_len))
            return _block_tail_x3(rt, o2, att, self.norm2, self.mlp, fused_any_rows=_RTSA_MLP_FUSED and att.dim in (128, 256))
        if self.use_layer_scale or (self.training and self.drop_path.drop_prob > 0.0):
            bid = plan.relay_cloud()
            rt = rt + self.drop_path(self.gamma1 * self.rt_attention(_ln(rt, self.norm1), plan), bid, plan.B)
            return rt + self.drop_path(self.gamma2 * self.mlp(_ln(rt, self.norm2)), bid, plan.B)
        att = self.rt_attention
        if _x6_path(rt, att.qkv, att.proj, self.mlp.fc1, self.mlp.fc2) and rt.shape[-1] in ops._LN_CHANNELS:
            o = ops.relay_attention(att.qkv(_ln(rt, self.norm1)), plan.seq_rows, plan.seq_off, plan.B, att.num_heads,
                                    plan.max_seq_len)
            return _block_tail_x6(rt, o, att.proj, self.norm2, self.mlp)
        rt, h = _add_ln(rt, self.rt_attention(_ln(rt, self.norm1), plan), self.norm2)
        return rt + self.mlp(h)


class RelayTokenInitialiser(nn.Module):
    """models/hotformerloc_backbone.py:305-363"""

    def __init__(self, dim, patch_size, conv_norm='layernorm', use_cpe=False, xcpe=False):
        super().__init__()
        self.patch_size = patch_size
        self.use_cpe = use_cpe
        if use_cpe:
            self.cpe = CPE(dim, conv_norm, xcpe)

    def forward(self, x, plan: WindowPlan, depth: int):
        if self.use_cpe:
            x = self.cpe(x, plan, depth, residual=False)
        if _grad_path(x):
            return ag.relay_token_init(x, plan.meta[depth], plan.n_windows[depth], self.patch_size)
        return ops.relay_token_init(x, plan.meta[depth], plan.n_windows[depth], self.patch_size)


class ADaPE(nn.Module):
    """models/layers/octformer_layers.py:177-210"""

    def __init__(self, dim, mode='cov'):
        super().__init__()
        self.mlp = MLP({'pos': 3, 'var': 6, 'cov': 9}[mode], dim, dim)

    def forward(self, plan: WindowPlan, depth: int):
        return self.mlp(plan.window_stats[depth])


class HOTFormerStage(nn.Module):
    """models/hotformerloc_backbone.py:366-635: one channel width for all pyramid levels (shipped configs), or one per level
    with linear projections between each level's width and the relay-token width (the widest level)."""

    def __init__(self, channels, num_heads, num_blocks, num_pyramid_levels, patch_size,
                 disable_RPE=False, ADaPE_mode=None, conv_norm='layernorm', drop_path=0.0, grad_checkpoint=False,
                 dilation=4, disable_rt=False, layer_scale=None, xcpe=False, rt_propagation=False,
                 rt_propagation_scale=None):
        super().__init__()
        self.grad_checkpoint = grad_checkpoint
        self.disable_rt = disable_rt
        channels, num_heads = list(channels), list(num_heads)
        self.use_projections = len(channels) != 1 and not disable_rt          # :384-400
        if len(channels) == 1:
            channels = channels * num_pyramid_levels
        if len(num_heads) == 1:
            num_heads = num_heads * num_pyramid_levels
        assert len(channels) == num_pyramid_levels, 'Invalid num channels specified'
        assert len(num_heads) == num_pyramid_levels, 'Invalid num heads specified'
        self.channels, self.num_heads = channels, num_heads
        Cm = self.max_rt_channels = max(channels)
        Hm = self.max_rt_num_heads = num_heads[channels.index(Cm)]
        self.num_pyramid_levels, self.num_blocks = num_pyramid_levels, num_blocks
        self.use_ADaPE = ADaPE_mode is not None
        dp = drop_path if isinstance(drop_path, (list, tuple)) else [drop_path] * num_blocks
        L = num_pyramid_levels
        if disable_rt:
            # ablation without relay tokens (hotformerloc_backbone.py:396-397,440-458,477): plain local-attention
            # blocks with the dilation re-enabled, no RTSA, no relay-token initialiser
            self.hosa_blocks = nn.ModuleList([
                nn.ModuleList([OctFormerBlock(channels[j], num_heads[j], patch_size, 1 if i % 2 == 0 else dilation,
                                              disable_RPE, conv_norm, dp[i], layer_scale, xcpe)
                               for i in range(num_blocks)]) for j in range(L)])
        else:
            self.hosa_blocks = nn.ModuleList([
                nn.ModuleList([HOTFormerBlock(channels[j], num_heads[j], patch_size, disable_RPE, conv_norm, dp[i],
                                              layer_scale, xcpe, rt_propagation, rt_propagation_scale,
                                              last=(i == num_blocks - 1))
                               for i in range(num_blocks)]) for j in range(L)])
            if self.use_projections:                                          # :406-408,461-475 (registration order)
                self.up_projections = nn.ModuleList([
                    nn.ModuleList([nn.Linear(channels[j], Cm) for _ in range(num_blocks)]) for j in range(L)])
                self.down_projections = nn.ModuleList([
                    nn.ModuleList([nn.Linear(Cm, channels[j]) for _ in range(num_blocks)]) for j in range(L)])
                self.init_up_projections = nn.ModuleList([nn.Linear(channels[j], Cm) for j in range(L)])
            self.rtsa_blocks = nn.ModuleList([RelayTokenTransformerBlock(Cm, Hm, dp[i], layer_scale)
                                              for i in range(num_blocks)])
            if self.use_projections:
                self.relay_tokeniser = nn.ModuleList([
                    RelayTokenInitialiser(channels[j], patch_size, conv_norm, use_cpe=not self.use_ADaPE, xcpe=xcpe)
                    for j in range(L)])
            else:
                self.relay_tokeniser = RelayTokenInitialiser(Cm, patch_size, conv_norm,
                                                             use_cpe=not self.use_ADaPE, xcpe=xcpe)
            if self.use_ADaPE:
                self.rt_adape = ADaPE(Cm, ADaPE_mode)
                if self.use_projections:
                    self.rt_adape_projections = nn.ModuleList([nn.Linear(Cm, channels[j]) for j in range(L)])
        self.downsamples = nn.ModuleList([Downsample(channels[j], channels[j + 1], conv_norm)
                                          for j in range(L - 1)])
        self._streams = None

    def _side_streams(self, device):
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=device) for _ in range(self.num_pyramid_levels - 1)]
        return self._streams

    def _rtsa_stream(self, device):
        if self.__dict__.get('_rtsa_st') is None:
            # high priority: RTSA is a chain of eight tiny launches that must slip in between the chip-filling kernels of the
            # token-row phase; at equal priority each of them queues behind a full round of workgroups
            self.__dict__['_rtsa_st'] = torch.cuda.Stream(device=device, priority=-1)
        return self.__dict__['_rtsa_st']

    def _forward_without_relay_tokens(self, data, plan: WindowPlan, depths):
        feats = {depths[0]: data}
        for j, d in enumerate(depths[:-1]):
            feats[d - 1] = self.downsamples[j](feats[d], plan.octree, d)
        ckpt = _use_checkpoint(self, sum(f.shape[0] * f.shape[1] for f in feats.values()), self.num_blocks, data.device)
        for i in range(self.num_blocks):
            for j, d in enumerate(depths):
                blk = self.hosa_blocks[j][i]
                feats[d] = _checkpoint_block(blk, feats[d], plan, d) if ckpt else blk(feats[d], plan, d)
        return feats, {d: None for d in depths}

    def forward(self, data, plan: WindowPlan, depth: int):
        depths = [depth - j for j in range(self.num_pyramid_levels)]
        if self.disable_rt:
            return self._forward_without_relay_tokens(data, plan, depths)
        octree = plan.octree
        feats = {depths[0]: data}
        bufs: Dict[int, torch.Tensor] = {}
        proj = self.use_projections
        for j, d in enumerate(depths):                                  # init_pyramid_feats, 540-572
            tokeniser = self.relay_tokeniser[j] if proj else self.relay_tokeniser
            rt = tokeniser(feats[d], plan, d)
            if self.use_ADaPE:
                pe = self.rt_adape(plan, d)
                rt = rt + (self.rt_adape_projections[j](pe) if proj else pe)
            bufs[d] = torch.cat([feats[d], rt], 0)
            if j < self.num_pyramid_levels - 1:
                feats[d - 1] = self.downsamples[j](feats[d], octree, d)
        nts = [plan.n_tokens[d] for d in depths]
        # relay rows as RTSA sees them: each level's own rows, or (per-level widths) their projection to the widest level
        rts = {d: (self.init_up_projections[j](bufs[d][nt:]) if proj else bufs[d][nt:])          # 585-591
               for j, (d, nt) in enumerate(zip(depths, nts))}
        ckpt = _use_checkpoint(self, sum(b.shape[0] * b.shape[1] for b in bufs.values()), self.num_blocks, data.device)

        def hosa(j, d, i, buf, fresh_d):
            """down-projection -> H-OSA block -> up-projection of one level (610-630); returns (buffer, relay rows for RTSA)"""
            blk = self.hosa_blocks[j][i]
            rin = self.down_projections[j][i](fresh_d) if proj else fresh_d
            out = _checkpoint_block(blk, buf, plan, d, rin) if ckpt else blk(buf, plan, d, rin)
            nt = plan.n_tokens[d]
            return out, (self.up_projections[j][i](out[nt:]) if proj else out[nt:])

        early = _EARLY_PHASE and _PYRAMID_STREAMS and not _grad_path(data) and data.is_cuda and not ckpt
        if early and not _ATTN_WS_EARLY and self._finest_level_fuses_attention(bufs[depths[0]], plan, depths[0]):
            # The finest level's LN1 -> qkv -> window attention is ONE launch there (csrc/attn_ws.hip) and needs the relay
            # rows: nothing but its CPE could run beside the relay-token block, and the persistent launch starves the coarse
            # levels' early phases.  RTSA first, then the levels side by side: +2-4 % on three boxes
            # (profiles/r05_w_ab_attn_ws_default.log); with the early phases kept the one-launch form gains nothing.
            early = False
        return self._iterations(data, plan, depths, bufs, rts, nts, proj, ckpt, early, hosa)

    def _finest_level_fuses_attention(self, buf, plan: WindowPlan, depth: int) -> bool:
        """The conditions under which _native_block_call gives the finest level's blocks the one-launch attention branch
        (without preparing a call)."""
        blk = self.hosa_blocks[0][0]
        att = blk.attention
        nt = plan.n_tokens[depth]
        if not (_NATIVE_BLOCK and _GEMM_MODE == 'x3' and _ATTN_F16 and _split_path(buf) and ops.KernelTimer.active is None
                and not blk.use_layer_scale and not _drops(blk) and not blk.cpe.xcpe and buf.dtype == torch.float32
                and _attn_f16_ok(buf.shape[0], att, depth)):
            return False
        return _attn_ws_wanted(att, buf.shape[0], nt, plan.n_windows[depth], depth) and _qkv_pack(att, nt) is not None

    def _iterations(self, data, plan, depths, bufs, rts, nts, proj, ckpt, early, hosa):
        done = None          # early schedule: per-level end-of-iteration events of the previous iteration
        first = None         # ... and the buffers the schedule started from (allocated on the main stream, read by the others)
        for i in range(self.num_blocks):                                # 593-633
            if early:
                # RTSA of iteration i only feeds the relay rows: what a block does with its TOKEN rows before the window
                # attention (CPE, LN1, qkv projection: a third of the block) does not wait for it.  Every level issues that
                # part on its own stream first, RTSA runs beside it on a stream of its own, the rest of the block follows
                # once both are done -- the ~120 us chain of eight tiny RTSA launches leaves the critical path.
                main = torch.cuda.current_stream()
                side = [main] * (len(depths) - 1) if _SERIAL_STREAMS else self._side_streams(data.device)
                rs = main if (_SERIAL_STREAMS or not _RTSA_STREAM) else self._rtsa_stream(data.device)
                small = [not (j == 0 or bufs[d].shape[0] > _SIDE_STREAM_MAX_ROWS) for j, d in enumerate(depths)]
                sts = [side[j - 1] if small[j] else main for j in range(len(depths))]
                # issue order = critical path first (the host runs only just ahead of the GPU here): the finest level's
                # token phase, RTSA, then the small levels
                order = sorted(range(len(depths)), key=lambda j: -bufs[depths[j]].shape[0])
                # Who waits for whom.  A level's block i needs that level's block i - 1 (stream order) and, for its relay rows,
                # RTSA i; RTSA i needs every level's block i - 1.  Nothing else: in particular the finest level's CPE / LN1 /
                # qkv of iteration i do not wait for the coarse levels' MLP launches of iteration i - 1, which run (starved)
                # beside and after the finest level's chip-filling fused MLP -- with a join of all streams at the end of every
                # iteration (HFL_ITER_JOIN=1, rounds 2-3) the finest level's queue stood idle ~80 us per iteration there
                # (kernel trace, profiles/r04_phases_iteration_timeline_join.log).
                join = _ITER_JOIN or done is None
                ev0 = main.record_event() if join else None
                if first is None:
                    first = (dict(bufs), dict(rts))
                calls = {}

                def phase1(j):
                    d = depths[j]
                    if join and sts[j] is not main:
                        sts[j].wait_event(ev0)
                    with torch.cuda.stream(sts[j]):
                        calls[d] = _native_block_call(self.hosa_blocks[j][i], bufs[d], plan, d)
                        if calls[d] is not None:
                            calls[d].run(1)

                phase1(order[0])
                if join:
                    rs.wait_event(ev0)
                else:
                    for ev in done:
                        rs.wait_event(ev)
                with torch.cuda.stream(rs):
                    rt_all = self.rtsa_blocks[i].forward_parts([rts[d] for d in depths], plan)
                    ev_rt = rs.record_event()
                for j in order[1:]:
                    phase1(j)
                fresh = {d: rt_all[plan.rt_offset[d]:plan.rt_offset[d] + plan.n_windows[d]] for d in depths}
                old = (dict(bufs), dict(rts))        # buffers other streams still read stay alive until the join
                # The levels on SIDE streams (the small ones: 2 k and 14 k rows in the bench) send their window attention out as
                # ONE launch: their windows are one or two per workgroup, latency-bound launches of 14 and 21 us alone.  The
                # join is between those side streams only.  (Measured alternatives: all three levels in one launch joined
                # on the main stream -3.5 % of the step; the small levels back to back on one side stream -3 %.)
                # (blocks on the one-launch LN1 -> qkv -> attention path have no separate attention launch to merge)
                group = [j for j in order if small[j] and calls[depths[j]] is not None
                         and not calls[depths[j]].weights.fuse_attention & 2] if _MERGED_ATTN else []
                if len(group) < 2:
                    group = []
                outs = {}

                def relay_in(j):
                    return self.down_projections[j][i](fresh[depths[j]]) if proj else fresh[depths[j]]

                for j in order:
                    d = depths[j]
                    sts[j].wait_event(ev_rt)
                    with torch.cuda.stream(sts[j]):
                        if j in group:
                            calls[d].run(3, relay_in(j))                      # relay rows in, their LN1 / qkv
                        elif calls[d] is not None:
                            outs[j] = self.hosa_blocks[j][i]._tail(calls[d].run(2, relay_in(j)), plan, d)
                        else:
                            outs[j] = self.hosa_blocks[j][i](bufs[d], plan, d, relay_in(j))
                if group:
                    lead = sts[group[0]]
                    for j in group[1:]:
                        lead.wait_event(sts[j].record_event())
                    with torch.cuda.stream(lead):
                        ops.block_attention_multi([calls[depths[j]] for j in group])
                        ev_att = lead.record_event()
                    for j in group:
                        if sts[j] is not lead:
                            sts[j].wait_event(ev_att)
                        with torch.cuda.stream(sts[j]):
                            outs[j] = self.hosa_blocks[j][i]._tail(calls[depths[j]].run(4), plan, depths[j])      # proj + MLP
                for j in order:
                    with torch.cuda.stream(sts[j]):
                        bufs[depths[j]] = outs[j]
                        rts[depths[j]] = self.up_projections[j][i](outs[j][nts[j]:]) if proj else outs[j][nts[j]:]
                if _ITER_JOIN or i + 1 == self.num_blocks:
                    for j in range(len(depths)):
                        if sts[j] is not main:
                            main.wait_stream(sts[j])
                else:
                    done = [st.record_event() for st in dict.fromkeys(sts)]
                del calls, old, fresh, rt_all
                continue
            call0 = None
            if (_CPE_FIRST and _PYRAMID_STREAMS and not _SERIAL_STREAMS and not ckpt and not _grad_path(data) and data.is_cuda
                    and not proj):
                # Probe (off): the finest level's CPE, which reads token rows only, issued BEFORE the relay-token block on the same
                # stream (no event hop), so that it runs alone instead of behind the coarse levels' forked launches (kernel
                # timeline profiles/r06_x_phases_iteration_5.log: 131 us there, 44 us alone).  Measured -1.9 % on the headline,
                # Oxford -0.4 % (profiles/r06_aa_ab_cpe_first.log): the relay-token block then starts 44 us later and everything
                # that waits for it with it.
                call0 = _native_block_call(self.hosa_blocks[0][i], bufs[depths[0]], plan, depths[0])
                if call0 is not None:
                    call0.run(1)
            if ckpt:                                                    # 596-601
                rt_all = _checkpoint_block(self.rtsa_blocks[i], torch.cat([rts[d] for d in depths], 0), plan)
            else:
                rt_all = self.rtsa_blocks[i].forward_parts([rts[d] for d in depths], plan)
            fresh = {d: rt_all[plan.rt_offset[d]:plan.rt_offset[d] + plan.n_windows[d]] for d in depths}
            if _PYRAMID_STREAMS and not _SERIAL_STREAMS and not _grad_path(data) and data.is_cuda:
                # the three depths are independent inside an iteration (the reference runs them on
                # three CUDA streams too, hotformerloc_backbone.py:604-633): the coarse depths'
                # small GEMMs and kernels overlap the fine depth's.  Fork/join discipline: side
                # streams wait for the main stream's event, the main stream waits for theirs.
                # Only SMALL depths go to a side stream (_SIDE_STREAM_MAX_ROWS): they are the ones
                # that cannot fill 256 CUs, and hipBLASLt's stream-K GEMMs (workgroups that wait on
                # their peers' partial tiles) dead-lock when several chip-filling ones share the CUs.
                main = torch.cuda.current_stream()
                side = self._side_streams(data.device)
                keep = []         # inputs allocated on the main stream stay alive until the join
                used = []
                for j, d in enumerate(depths):
                    if j == 0 or bufs[d].shape[0] > _SIDE_STREAM_MAX_ROWS:
                        continue
                    side[j - 1].wait_stream(main)
                    keep.append((bufs[d], rt_all))
                    with torch.cuda.stream(side[j - 1]):
                        bufs[d], rts[d] = hosa(j, d, i, bufs[d], fresh[d])
                    used.append(j)
                for j, d in enumerate(depths):
                    if j == 0 and call0 is not None:            # the rest of the block whose CPE went out before RTSA
                        out = self.hosa_blocks[0][i]._tail(call0.run(2, fresh[d]), plan, d)
                        bufs[d], rts[d] = out, out[nts[0]:]
                    elif j not in used:
                        bufs[d], rts[d] = hosa(j, d, i, bufs[d], fresh[d])
                for j in used:
                    main.wait_stream(side[j - 1])
                del keep, call0
            else:
                for j, d in enumerate(depths):
                    bufs[d], rts[d] = hosa(j, d, i, bufs[d], fresh[d])
        local = {d: bufs[d][:nt] for d, nt in zip(depths, nts)}
        return local, rts


class HOTFormerBase(nn.Module):
    """models/hotformerloc_backbone.py:638-723"""

    def __init__(self, in_channels, channels, num_blocks, num_heads, num_pyramid_levels,
                 num_octf_levels, patch_size, dilation, stem_down, ADaPE_mode, disable_RPE, conv_norm,
                 drop_path=0.0, grad_checkpoint=False, disable_rt=False, layer_scale=None, xcpe=False,
                 rt_propagation=False, rt_propagation_scale=None, downsample_input_embeddings=True):
        super().__init__()
        self.downsample_input_embeddings = downsample_input_embeddings
        # stochastic depth per block (hotformerloc_backbone.py:669-700)
        drop_ratio = torch.linspace(0, drop_path, sum(num_blocks)).tolist()
        self.patch_size, self.dilation = patch_size, dilation
        self.num_pyramid_levels, self.num_octf_levels = num_pyramid_levels, num_octf_levels
        self.num_stages = num_octf_levels + num_pyramid_levels
        self.stem_down = stem_down
        self.ADaPE_mode = ADaPE_mode
        if num_heads is None:
            num_heads = [c // 16 for c in channels]
        self.patch_embed = PatchEmbed(in_channels, channels[0], stem_down, conv_norm, downsample_input_embeddings)
        self.octf_stage = nn.ModuleList([
            OctFormerStage(channels[i], num_heads[i], patch_size, dilation, num_blocks[i], disable_RPE,
                           conv_norm, drop_ratio[sum(num_blocks[:i]):sum(num_blocks[:i + 1])], grad_checkpoint,
                           layer_scale, xcpe)
            for i in range(num_octf_levels)])
        self.downsample = nn.ModuleList([Downsample(channels[i], channels[i + 1], conv_norm)
                                         for i in range(num_octf_levels)])
        self.hotf_stage = HOTFormerStage(list(channels[num_octf_levels:]),
                                         list(num_heads[num_octf_levels:]), num_blocks[-1],
                                         num_pyramid_levels, patch_size, disable_RPE, ADaPE_mode,
                                         conv_norm, drop_ratio[sum(num_blocks[:-1]):sum(num_blocks)],
                                         grad_checkpoint, dilation, disable_rt, layer_scale, xcpe, rt_propagation,
                                         rt_propagation_scale)

    def forward(self, data, octree, depth):
        # the plan depends on the octree only: built first, so that its host work hides the round trip of the tap counts
        # that `construct_all_neigh()` started (the stem convolutions below are the first to need them)
        top = depth - self.stem_down if self.downsample_input_embeddings else depth         # :707-708

        def make_plan():
            return WindowPlan.for_octree(octree, self.patch_size, self.dilation, max_depth=top,
                                         start_depth=top - self.num_stages + 1,
                                         num_pyramid_levels=self.num_pyramid_levels,
                                         num_octf_levels=self.num_octf_levels, adape_mode=self.ADaPE_mode)

        if _PLAN_LATE and not _grad_path() and data.is_cuda:
            # inference: the stem first (it only needs the tap lists, whose counts the host has to wait for anyway), THEN the
            # plan: its ~0.25 ms of host work runs while the GPU executes the stem's 0.7 ms instead of in front of it with
            # the GPU idle (kernel trace of a step's first millisecond, profiles/r04_phases_stem_head_end_of_round.log)
            data = self.patch_embed(data, octree, depth)
            plan = make_plan()
        else:
            plan = make_plan()
            data = self.patch_embed(data, octree, depth)
        depth = top
        for i in range(self.num_octf_levels):
            data = self.octf_stage[i](data, plan, depth)
            data = self.downsample[i](data, octree, depth)
            depth -= 1
        local, relay = self.hotf_stage(data, plan, depth)
        return local, relay, plan


class HOTFormer(nn.Module):
    """models/hotformerloc_backbone.py:726-849 (init: 817-843)"""

    def __init__(self, in_channels, channels, num_blocks, num_heads, num_pyramid_levels=3,
                 num_octf_levels=1, patch_size=32, dilation=4, drop_path=0.5, stem_down=2,
                 ADaPE_mode=None, disable_RPE=False, conv_norm='layernorm',
                 qkv_init=('trunc_normal', 0.02), grad_checkpoint=False, disable_rt=False, layer_scale=None,
                 xcpe=False, rt_size=1, rt_propagation=False, rt_propagation_scale=None,
                 downsample_input_embeddings=True):
        super().__init__()
        if rt_size != 1:
            # the reference's own model cannot run this option: RelayTokenInitialiser views the windows as (-1, K // G, C)
            # against a (windows, K) mask and raises (models/hotformerloc_backbone.py:354-357, "TODO: Make this work with
            # rt_size > 1"); tests/test_variants.py::test_ct_size_other_than_one_is_rejected_like_the_reference
            raise NotImplementedError('ct_size != 1: the reference model itself fails on it '
                                      '(models/hotformerloc_backbone.py:354-357)')
        self.backbone = HOTFormerBase(in_channels, list(channels), list(num_blocks),
                                      None if num_heads is None else list(num_heads),
                                      num_pyramid_levels, num_octf_levels, patch_size, dilation,
                                      stem_down, ADaPE_mode, disable_RPE, conv_norm, drop_path, grad_checkpoint,
                                      disable_rt, layer_scale, xcpe, rt_propagation, rt_propagation_scale,
                                      downsample_input_embeddings)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        if qkv_init[0] == 'trunc_normal':
            for name, m in self.named_modules():
                if 'qkv' in name and isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=qkv_init[1])

    def forward(self, data, octree, depth):
        return self.backbone(data, octree, depth)


# -------------------------------------------------------------------------- pooling
class AdaptivePooling(nn.Module):
    """models/layers/salsa.py:12-55: learned queries attend over one cloud's tokens."""

    def __init__(self, feature_dim, k_pooled_tokens):
        super().__init__()
        self.query = nn.Parameter(torch.randn(k_pooled_tokens, feature_dim))
        self.scale = feature_dim ** -0.5

    def forward(self, x, plan: WindowPlan, depth: int, out=None):
        """x (N_t, C) ragged over clouds -> (B, k, C) (`out`: where to, e.g. this level's slice of the token matrix)."""
        if _grad_path(x):
            return ag.attentional_pooling_torch(x, self.query, plan, depth, self.scale)
        if _ATTN_POOL and _GEMM_MODE == 'x3' and _split_path(x) and x.dtype == torch.float32 and ops.attn_pool_ok(x.shape[1]):
            # scores -> softmax over the cloud's rows -> weighted sum in one launch, nothing padded (csrc/attn_pool.hip)
            return ops.attn_pool(x, plan.cloud_off[depth], self.query, plan.B, self.scale, out=out)
        scores = torch.mm(x, self.query.t())                              # (N_t, k)
        ops.segment_softmax_(scores, plan.cloud_off[depth], plan.B, self.scale)
        nmax = plan.pad_index[depth].numel() // plan.B
        if scores.shape[1] % 4 == 0:                     # per-cloud zero-padded copies in one pass each (hfl_pad_rows)
            xp = ops.pad_rows(x, plan.cloud_off[depth], plan.B, nmax)
            pp = ops.pad_rows(scores, plan.cloud_off[depth], plan.B, nmax)
        else:
            idx = plan.pad_index[depth]
            zero = x.new_zeros(1, x.shape[1])
            xp = torch.cat([x, zero], 0).index_select(0, idx).view(plan.B, -1, x.shape[1])
            pp = torch.cat([scores, scores.new_zeros(1, scores.shape[1])], 0) \
                .index_select(0, idx).view(plan.B, -1, scores.shape[1])
        return torch.bmm(pp.transpose(1, 2), xp)


class FeatureMixerLayer(nn.Module):
    """models/layers/salsa.py:58-75"""

    def __init__(self, in_dim, mlp_ratio=1):
        super().__init__()
        self.mix = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, int(in_dim * mlp_ratio)),
                                 nn.GELU(), nn.Linear(int(in_dim * mlp_ratio), in_dim))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        return x + self.mix(x)


_MIXER_FUSED = _knob('HFL_MIXER_FUSED', '1') != '0'


def _mixer_pack(fc1: nn.Linear, fc2: nn.Linear):
    """`ops.mlp_fused_pack` image of a Mixer layer's two Linears, cached per parameter pair like `_mlp_pack`."""
    w1, w2 = fc1.weight, fc2.weight
    key = ('mixer', id(w1), id(w2))
    hit = _W3_CACHE.get(key)
    stamp = (w1._version, w2._version, w1.data_ptr(), w2.data_ptr())
    if hit is None or hit[0]() is not w1 or hit[1]() is not w2 or hit[2] != stamp:
        if hit is None or hit[0]() is not w1:
            weakref.finalize(w1, _W3_CACHE.pop, key, None)
        hit = (weakref.ref(w1), weakref.ref(w2), stamp, ops.mlp_fused_pack(w1, w2))
        _W3_CACHE[key] = hit
    return hit[3]


def _mixer_layer_fits(m) -> bool:
    ln, fc1, act, fc2 = m.mix
    return (isinstance(ln, nn.LayerNorm) and isinstance(act, nn.GELU) and fc1.in_features % 128 == 0
            and fc1.out_features % 128 == 0 and fc2.out_features % 128 == 0 and fc1.bias is not None and fc2.bias is not None)


class Mixer(nn.Module):
    """models/layers/salsa.py:78-111"""

    def __init__(self, k_input_tokens, k_output_tokens, in_d, mix_depth, mlp_ratio, out_d):
        super().__init__()
        self.mix = nn.Sequential(*[FeatureMixerLayer(in_d, mlp_ratio) for _ in range(mix_depth)])
        self.row_proj = nn.Linear(in_d, out_d)
        self.channel_proj = nn.Linear(k_input_tokens, k_output_tokens)

    def forward(self, x):
        if _split_path(x) and x.shape[-1] % 128 == 0 and all(_mixer_layer_fits(m) for m in self.mix):
            # inference: every FeatureMixerLayer as LayerNorm -> split2, fc1 + GELU, fc2 + residual on the hand-written split GEMM
            # (three launches instead of LayerNorm, addmm, GELU, addmm, add: the (B k, C) token matrix is 8 k rows, every one of
            # these launches is latency-bound and the host was issuing them into an idle GPU)
            b, k, c = x.shape
            x2 = x.reshape(b * k, c)
            for m in self.mix:
                ln, fc1, _, fc2 = m.mix
                if _MIXER_FUSED and _GEMM_MODE == 'x3' and ops.mlp_fused_shape_ok(c, fc1.out_features):
                    # the whole layer as the block MLP's launch (hidden = C): one launch (+ the hidden split's reduce) for three
                    x2 = ops.ln_mlp_fused(x2.contiguous(), ln.weight, ln.bias, ln.eps, _mixer_pack(fc1, fc2), fc1.bias, fc2.bias)
                    continue
                h2 = ops.layer_norm_split2(x2, ln.weight, ln.bias, ln.eps)
                g2 = ops.linear_x3(h2, _w2(fc1), bias=fc1.bias, gelu_split_out=True)
                x2 = ops.linear_x3(g2, _w2(fc2), bias=fc2.bias, residual=x2)
            x = x2.view(b, k, c)
        elif all(_mixer_layer_fits(m) for m in self.mix) and _x6_path(x, *[l for m in self.mix for l in (m.mix[1], m.mix[3])]):
            # matched precision: LayerNorm, fc1 + bias + GELU, fc2 + bias + residual on hfl_linear_x6 (three launches per layer)
            b, k, c = x.shape
            x2 = x.reshape(b * k, c).contiguous()
            for m in self.mix:
                ln, fc1, _, fc2 = m.mix
                h = ops.layer_norm(x2, ln.weight, ln.bias, ln.eps)
                g = ops.linear_x6(h, _w6(fc1), bias=fc1.bias, gelu=True)
                x2 = ops.linear_x6(g, _w6(fc2), bias=fc2.bias, residual=x2)
            x = x2.view(b, k, c)
        else:
            x = self.mix(x)
        if (_MIXER_FUSED and _GEMM_MODE == 'x3' and x.is_cuda and not _grad_path() and x.dtype == torch.float32
                and x.shape[-1] % 4 == 0 and self.row_proj.out_features <= 8 and self.channel_proj.bias is not None
                and self.row_proj.bias is not None and x.shape[1] * self.row_proj.out_features <= 16000):
            # channel_proj, row_proj and the flatten as one small launch (row_proj first: the maps commute)
            return ops.mixer_tail(x, self.channel_proj.weight, self.channel_proj.bias, self.row_proj.weight, self.row_proj.bias)
        x = self.channel_proj(x.permute(0, 2, 1)).permute(0, 2, 1)
        return self.row_proj(x).flatten(1)


class PyramidAttnPoolWrapper(nn.Module):
    """models/layers/pooling.py:106-233 (aggregator='mixer')"""

    def __init__(self, feature_size, output_dim, channels, num_pyramid_levels, k_pooled_tokens,
                 mlp_ratio=1, mix_depth=4):
        super().__init__()
        channels = list(channels)
        self.use_projections = len(channels) != 1                               # pooling.py:123-129
        if len(channels) == 1:
            channels = channels * num_pyramid_levels
        assert len(channels) == num_pyramid_levels, 'Incorrect num channels'
        assert len(k_pooled_tokens) == num_pyramid_levels, \
            'k_pooled_tokens must be list of k for each pyramid level'
        self.k_pooled_tokens = list(k_pooled_tokens)
        total = sum(k_pooled_tokens)
        self.attpool = nn.ModuleList([AdaptivePooling(channels[j], k) for j, k in enumerate(k_pooled_tokens)])
        if self.use_projections:                                                # pooling.py:143-152
            self.local_projections = nn.ModuleList([nn.Linear(c, feature_size) if c != feature_size else nn.Identity()
                                                    for c in channels])
        k_out = total // 4
        out_d = output_dim // k_out
        assert k_out * out_d == output_dim, \
            f'Invalid k for k_pooled_tokens: {k_pooled_tokens}, not compatible with output dim {output_dim}'
        self.descriptor_extractor = Mixer(total, k_out, feature_size, mix_depth, mlp_ratio, out_d)

    def forward(self, local_feat_dict, plan: WindowPlan, depth=None):
        feats = list(local_feat_dict.items())
        if (not self.use_projections and not _grad_path() and feats and feats[0][1].is_cuda
                and all(f.shape[1] == feats[0][1].shape[1] and f.dtype == torch.float32 for _, f in feats)):
            # every level writes its k_j tokens straight into the (B, sum k, C) token matrix: no torch.cat
            all_t = torch.empty((plan.B, sum(self.k_pooled_tokens), feats[0][1].shape[1]), dtype=torch.float32,
                                device=feats[0][1].device)
            off = 0
            for j, (d, f) in enumerate(feats):
                k = self.k_pooled_tokens[j]
                t = self.attpool[j](f, plan, d, out=all_t[:, off:off + k])
                if t.data_ptr() != all_t[:, off:off + k].data_ptr():
                    all_t[:, off:off + k].copy_(t)
                off += k
            return self.descriptor_extractor(all_t)
        toks = []
        for j, d in enumerate(local_feat_dict.keys()):
            t = self.attpool[j](local_feat_dict[d], plan, d)
            toks.append(self.local_projections[j](t) if self.use_projections else t)
        return self.descriptor_extractor(torch.cat(toks, 1))


class OctGeM(nn.Module):
    """models/layers/pooling.py:18-40: generalised-mean pooling of the finest pyramid level over each cloud's
    non-empty nodes (`ocnn.nn.OctreeGlobalPool(nempty=True)` = per-cloud mean)."""

    def __init__(self, input_dim, p=3, eps=1e-6):
        super().__init__()
        self.input_dim = self.output_dim = input_dim
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps

    def forward(self, x, plan: WindowPlan, depth=None):
        if isinstance(x, dict):
            depth, x = max(x.items())
        temp = x.clamp(min=self.eps).pow(self.p)
        return _cloud_mean(temp, plan, depth).pow(1. / self.p)


class RelayTokenGeM(nn.Module):
    """models/layers/pooling.py:43-57 on a padded (B, N, C) tensor (every row counts, as in the reference)."""

    def __init__(self, input_dim, p=3, eps=1e-6):
        super().__init__()
        self.input_dim = self.output_dim = input_dim
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps

    def forward(self, x):
        return x.clamp(min=self.eps).pow(self.p).mean(dim=1).pow(1. / self.p)


class GatingContext(nn.Module):
    """models/layers/netvlad.py:83-112"""

    def __init__(self, dim, add_batch_norm=True):
        super().__init__()
        self.dim, self.add_batch_norm = dim, add_batch_norm
        self.gating_weights = nn.Parameter(torch.randn(dim, dim) / dim ** 0.5)
        if add_batch_norm:
            self.gating_biases = None
            self.bn1 = nn.BatchNorm1d(dim)
        else:
            self.gating_biases = nn.Parameter(torch.randn(dim) / dim ** 0.5)
            self.bn1 = None

    def forward(self, x):
        gates = torch.matmul(x, self.gating_weights)
        gates = self.bn1(gates) if self.add_batch_norm else gates + self.gating_biases
        return x * torch.sigmoid(gates)


class PyramidOctGeMWrapper(nn.Module):
    """models/layers/pooling.py:60-103: GeM per pyramid level (own exponent each), concatenated, Linear + BatchNorm,
    optional context gating."""

    def __init__(self, input_dim, output_dim, num_pyramid_levels, channels, p=3, eps=1e-6, gating=False,
                 add_batch_norm=True):
        super().__init__()
        assert num_pyramid_levels > 0, 'Minimum 1 pyramid layer'
        if len(channels) == 1:                                                  # pooling.py:66-70
            concat_dim = input_dim * num_pyramid_levels
        else:
            assert len(channels) == num_pyramid_levels, 'Incorrect num channels'
            concat_dim = sum(channels)
        self.input_dim, self.output_dim, self.num_pyramid_levels = input_dim, output_dim, num_pyramid_levels
        self.p = nn.Parameter(torch.ones(num_pyramid_levels) * p)
        self.eps, self.gating = eps, gating
        self.linear_bn = nn.Sequential(nn.Linear(concat_dim, output_dim, bias=False),
                                       nn.BatchNorm1d(input_dim))
        if gating:
            self.context_gating = GatingContext(output_dim, add_batch_norm=add_batch_norm)

    def forward(self, local_feat_dict, plan: WindowPlan, depth=None):
        desc = []
        for j, d in enumerate(local_feat_dict.keys()):
            temp = local_feat_dict[d].clamp(min=self.eps).pow(self.p[j])
            desc.append(_cloud_mean(temp, plan, d).pow(1. / self.p[j]))
        g = self.linear_bn(torch.cat(desc, dim=-1))
        return self.context_gating(g) if self.gating else g


class AttnPoolWrapper(nn.Module):
    """models/layers/pooling.py:235-305: attentional pooling of every cloud's multi-scale relay tokens (fine to coarse,
    `relay_token_utils.py:12-40`) to k tokens, then the MLP mixer or LayerNorm + MLP + GeM."""

    def __init__(self, feature_size=256, output_dim=256, k_pooled_tokens=64, mlp_ratio=1, aggregator='mixer',
                 mix_depth=4):
        super().__init__()
        assert isinstance(k_pooled_tokens, int), 'Only 1 value allowed for k_pooled_tokens when using relay tokens'
        self.feature_size, self.output_dim, self.k_pooled_tokens = feature_size, output_dim, k_pooled_tokens
        self.aggregator = aggregator
        self.attpool = AdaptivePooling(feature_size, k_pooled_tokens)
        if aggregator.lower() == 'mixer':
            k_out = k_pooled_tokens // 4
            self.descriptor_extractor = Mixer(k_pooled_tokens, k_out, feature_size, mix_depth, mlp_ratio,
                                              output_dim // k_out)
        elif aggregator.lower() == 'gem':
            self.token_processor = nn.Sequential(nn.LayerNorm(feature_size),
                                                 MLP(feature_size, feature_size * mlp_ratio, output_dim))
            self.descriptor_extractor = RelayTokenGeM(input_dim=feature_size)
        else:
            raise NotImplementedError(f'No valid aggregator: {aggregator}')

    def forward(self, relay_token_dict, plan: WindowPlan, depth=None):
        rt_all = torch.cat([relay_token_dict[d] for d in plan.pyramid_depths], 0)      # rows as plan.rt_offset
        idx, valid = plan.relay_pad_index()                                            # (B, Rmax): cloud -> its rows
        x = rt_all[idx.clamp(min=0)] * valid.unsqueeze(-1).to(rt_all.dtype)            # zero padding rows
        # learned queries attend over the cloud's own tokens; padding keys get the reference's -1e3 additive mask
        scores = torch.matmul(self.attpool.query * self.attpool.scale, x.transpose(1, 2))          # (B, k, Rmax)
        scores = scores + (~valid).unsqueeze(1).to(scores.dtype) * -1e3
        tok = torch.matmul(torch.softmax(scores, dim=-1), x)                                      # (B, k, C)
        if self.aggregator.lower() != 'mixer':
            tok = tok + self.token_processor(tok)
        return self.descriptor_extractor(tok)


def _cloud_mean(x, plan: WindowPlan, depth: int):
    """(N_t, C) rows of one depth -> (B, C) mean over each cloud's rows (fixed summation order: padded gather)."""
    idx = plan.pad_index_for(depth)
    zero = x.new_zeros(1, x.shape[1])
    xp = torch.cat([x, zero], 0).index_select(0, idx).view(plan.B, -1, x.shape[1])
    cnt = plan.cloud_count(depth).clamp(min=1).to(x.dtype)
    return xp.sum(1) / cnt.unsqueeze(1)


class PoolingWrapper(nn.Module):
    """models/layers/pooling_wrapper.py:11-77"""

    def __init__(self, pool_method, in_dim, output_dim, num_pyramid_levels=None, channels=None,
                 k_pooled_tokens=None):
        super().__init__()
        self.pool_method, self.in_dim, self.output_dim = pool_method, in_dim, output_dim
        self.pooled_feats = 'local'             # flag if local feats or relay tokens are pooled
        if pool_method == 'OctGeM':
            assert in_dim == output_dim
            self.pooling = OctGeM(input_dim=in_dim)
        elif pool_method in ('PyramidOctGeM', 'PyramidOctGeMgc'):
            self.pooling = PyramidOctGeMWrapper(in_dim, output_dim, num_pyramid_levels, list(channels),
                                                gating=pool_method.endswith('gc'))
        elif pool_method == 'PyramidAttnPoolMixer':
            self.pooling = PyramidAttnPoolWrapper(in_dim, output_dim, list(channels), num_pyramid_levels,
                                                  k_pooled_tokens)
        elif pool_method in ('AttnPoolMixer', 'AttnPoolGeM'):
            self.pooled_feats = 'relaytokens'
            self.pooling = AttnPoolWrapper(in_dim, output_dim, k_pooled_tokens,
                                           aggregator='mixer' if pool_method == 'AttnPoolMixer' else 'GeM')
        elif pool_method == 'PyramidNetVLAD':
            raise NotImplementedError(f'Not implemented yet: {pool_method}')       # as the reference
        else:
            raise NotImplementedError('Unknown pooling method: {}'.format(pool_method))

    def forward(self, x, octree=None, depth=None):
        return self.pooling(x, octree, depth)


# ------------------------------------------------------------------------- wrapper
class HOTFormerLoc(nn.Module):
    """models/hotformerloc.py:18-82"""

    def __init__(self, backbone: nn.Module, pooling: PoolingWrapper, normalize_embeddings=False,
                 input_features='P'):
        super().__init__()
        if input_features != 'P':
            raise NotImplementedError("input_features=%r: every shipped config uses 'P'" % input_features)
        self.backbone = backbone
        self.pooling = pooling
        self.normalize_embeddings = normalize_embeddings
        self.input_features = input_features
        self.stats = {}

    def forward(self, batch):
        octree = batch['octree']
        if octree.device.type != 'cuda':
            raise RuntimeError('HOTFormerLoc (MI355X build) needs the octree on a GPU; '
                               'call to_device(batch, "cuda") first -- there is no CPU path')
        if _MAIN_HI and not _grad_path() and not _SERIAL_STREAMS:
            # probe (HFL_MAIN_HI=1): the whole inference forward on a HIGH-priority stream, so that the finest pyramid level's
            # chain (which stays on it) wins the CUs against the coarse levels' side streams (normal priority)
            hi = self.__dict__.get('_hi_stream')
            if hi is None or hi.device != octree.device:
                hi = self.__dict__['_hi_stream'] = torch.cuda.Stream(device=octree.device, priority=-1)
            cur = torch.cuda.current_stream(octree.device)
            hi.wait_stream(cur)
            with torch.cuda.stream(hi):
                out = self._forward(batch)
            cur.wait_stream(hi)
            out['global'].record_stream(cur)
            return out
        return self._forward(batch)

    def _forward(self, batch):
        octree = batch['octree']
        octree.construct_all_neigh()                     # no-op when misc/torch_utils.to_device did it
        data = octree.get_input_feature(self.input_features, nempty=True)
        armed = self.training and _DROP_POOL
        if armed:
            arm_drop_paths(self, int(octree.batch_size), data.device, data.dtype)
        try:
            local, relay, plan = self.backbone(data, octree, octree.depth)
        finally:
            if armed:
                disarm_drop_paths(self)
        if self.pooling.pooled_feats == 'local':
            x = local
        elif self.pooling.pooled_feats == 'relaytokens':
            x = relay
        else:
            raise ValueError(f"Invalid option for pooled features: '{self.pooling.pooled_feats}'")
        x = self.pooling(x, octree=plan)
        assert x.dim() == 2 and x.shape[1] == self.pooling.output_dim
        if self.normalize_embeddings:
            x = F.normalize(x, dim=1)
        return {'global': x}

    def print_info(self):
        n = sum(p.nelement() for p in self.parameters())
        print('Model class: HOTFormerLoc (MI355X build)')
        print(f'Total parameters: {n}')
        print(f'Pooling method: {self.pooling.pool_method}')
        print(f'Embedding normalization: {self.normalize_embeddings}')
