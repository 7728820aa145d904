"""ctypes binding of libhotformerloc_hip.so (the C-ABI in include/hotformerloc_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails,
this module raises.  Build it with `python -m hotformerloc_amd.build`
(or `__graft_entry__.build()`); it is loaded from `hotformerloc_amd/lib/` in-tree.
"""

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libhotformerloc_hip.so')

HFL_OCTREE_MAX_POINTS = 1 << 20
HFL_OCTREE_MAX_DEPTH = 10


class NativeLibraryError(RuntimeError):
    pass


class WindowAttnDesc(ctypes.Structure):
    """hfl_window_attn_desc"""
    _fields_ = [('n_tokens', c_int64), ('rt_row0', c_int64), ('n_windows', c_int32),
                ('patch_size', c_int32), ('dilation', c_int32), ('n_relay', c_int32),
                ('n_heads', c_int32), ('pos_bnd', c_int32), ('batch_size', c_int32),
                ('scale', c_float), ('depth', c_int32), ('rpe_expanded', c_void_p)]


class BlockWeights(ctypes.Structure):
    """hfl_block_weights"""
    _fields_ = [('channels', c_int64), ('eps', c_float), ('q_scale', c_float),
                ('cpe_weight', c_void_p), ('cpe_gamma', c_void_p), ('cpe_beta', c_void_p),
                ('norm1_gamma', c_void_p), ('norm1_beta', c_void_p), ('norm2_gamma', c_void_p), ('norm2_beta', c_void_p),
                ('qkv_w', c_void_p), ('proj_w', c_void_p), ('fc1_w', c_void_p), ('fc2_w', c_void_p),
                ('qkv_b', c_void_p), ('proj_b', c_void_p), ('fc1_b', c_void_p), ('fc2_b', c_void_p),
                ('rpe_table', c_void_p), ('mlp_pack', c_void_p), ('qkv_pack', c_void_p), ('fuse_attention', c_int32),
                ('rpe_tables3', c_void_p)]


class RelayBlockWeights(ctypes.Structure):
    """hfl_relay_block_weights"""
    _fields_ = [('channels', c_int64), ('n_heads', c_int32), ('eps', c_float),
                ('norm1_gamma', c_void_p), ('norm1_beta', c_void_p), ('norm2_gamma', c_void_p), ('norm2_beta', c_void_p),
                ('qkv_w', c_void_p), ('proj_w', c_void_p), ('fc1_w', c_void_p), ('fc2_w', c_void_p),
                ('qkv_b', c_void_p), ('proj_b', c_void_p), ('fc1_b', c_void_p), ('fc2_b', c_void_p), ('mlp_pack', c_void_p),
                ('qkv_pack', c_void_p)]


class RelayBlockIO(ctypes.Structure):
    """hfl_relay_block_io"""
    _fields_ = [('x_in', c_void_p), ('out', c_void_p), ('arena', c_void_p), ('seq_rows', c_void_p), ('seq_off', c_void_p),
                ('n_rows', c_int64), ('batch', c_int32), ('max_seq_len', c_int32), ('orphan_rows', c_void_p),
                ('n_orphans', c_int32), ('x_segments', c_void_p)]


class RowSegments(ctypes.Structure):
    """hfl_row_segments"""
    _fields_ = [('n', c_int32), ('ptr', c_void_p * 4), ('row0', c_int64 * 4)]


class BlockIO(ctypes.Structure):
    """hfl_block_io"""
    _fields_ = [('x_in', c_void_p), ('relay', c_void_p), ('out', c_void_p), ('arena', c_void_p),
                ('neigh', c_void_p), ('tok_meta', c_void_p), ('n_rows', c_int64), ('n_tokens', c_int64),
                ('phase', c_int32)]


# name -> (restype, argtypes): every symbol include/hotformerloc_hip.h declares
SIGNATURES = {
    'hfl_version': (c_int, []),
    'hfl_arch': (c_char_p, []),
    'hfl_dwconv_forward_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                            c_int64, c_int, c_void_p]),
    'hfl_dwconv_weight_backward_workspace': (c_int64, [c_int64, c_int64, c_int]),
    'hfl_dwconv_weight_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                           c_int64, c_int, c_void_p, c_void_p]),
    'hfl_inverse_neigh': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    'hfl_cpe_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_int64, c_int, c_float, c_int, c_void_p]),
    'hfl_cpe_forward_save': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_int64, c_int, c_float, c_int, c_void_p]),
    'hfl_dwconv_add': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    'hfl_octree_scratch_bytes': (c_int64, [c_int64, c_int, c_int, c_int]),
    'hfl_octree_build_clouds': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    'hfl_octree_merge': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    'hfl_octree_neigh': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                 c_int, c_void_p]),
    'hfl_token_meta': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'hfl_octree_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    'hfl_prepare_clouds': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'hfl_tap_lists_workspace': (c_int64, [c_int64, c_int]),
    'hfl_tap_lists': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    'hfl_tap_lists_multi': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'hfl_tap_tiles': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'hfl_pad_index': (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    'hfl_pad_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    'hfl_mixer_tail': (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p]),
    'hfl_attn_pool_ok': (c_int, [c_int]),
    'hfl_attn_pool_workspace': (c_int64, [c_int, c_int, c_int, c_int64]),
    'hfl_attn_pool': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_float, c_void_p,
                              c_int64, c_void_p]),
    'hfl_window_attention_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p,
                                         ctypes.POINTER(WindowAttnDesc), c_void_p]),
    'hfl_window_attention_fwd_ex': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            ctypes.POINTER(WindowAttnDesc), c_int, c_void_p]),
    'hfl_window_attention_fwd_multi': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'hfl_layer_norm_split3': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                      c_float, c_void_p]),
    'hfl_add_layer_norm_split3': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_int64, c_float, c_void_p]),
    'hfl_add_bias': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'hfl_bias_gelu_split3': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'hfl_split3': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'hfl_window_attention_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         ctypes.POINTER(WindowAttnDesc), c_void_p]),
    'hfl_window_attention_bwd_workspace': (c_int64, [ctypes.POINTER(WindowAttnDesc)]),
    'hfl_window_attention_bwd_det': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             ctypes.POINTER(WindowAttnDesc), c_void_p, c_void_p]),
    'hfl_window_attention_bwd_split2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                ctypes.POINTER(WindowAttnDesc), c_void_p, c_void_p]),
    'hfl_relay_attention_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_float, c_int, c_void_p]),
    'hfl_inverse_table': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    'hfl_octree_gather_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    'hfl_relay_token_init_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                         c_int64, c_void_p]),
    'hfl_linear_x3': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'hfl_linear_x6_grouped_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                     c_void_p]),
    'hfl_linear_x6_padded_k': (c_int64, [c_int64]),
    'hfl_linear_x6_pack': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'hfl_linear_x6': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    'hfl_layer_norm_split2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                      c_float, c_void_p]),
    'hfl_layer_norm_relu': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p]),
    'hfl_linear_x3_qkv': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    'hfl_window_attention_f16_ok': (c_int, [ctypes.POINTER(WindowAttnDesc), c_int64]),
    'hfl_split2': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'hfl_block_forward_x3_arena': (c_int64, [c_int64, c_int64]),
    'hfl_block_forward_x3': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'hfl_block_attention_x3_multi': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'hfl_relay_block_forward_x3_arena': (c_int64, [c_int64, c_int64]),
    'hfl_relay_block_forward_x3': (c_int, [c_void_p, c_void_p, c_void_p]),
    'hfl_linear_x3_grouped': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    'hfl_linear_x3_grouped_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                     c_int, c_void_p]),
    'hfl_split2_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'hfl_linear_x3_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'hfl_linear_x3_gelu_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'hfl_linear_x3_gelu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'hfl_tap_wgrad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                             c_void_p]),
    'hfl_slot_sum': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    'hfl_tap_wgrad_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                    c_int, c_void_p, c_void_p]),
    'hfl_wgrad_x3_workspace': (c_int64, [c_int64, c_int64, c_int64]),
    'hfl_wgrad_x3': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    'hfl_layer_norm_bwd_blocks': (c_int, [c_int64, c_int64]),
    'hfl_layer_norm_bwd_add': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                       c_float, c_void_p]),
    'hfl_layer_norm_bwd_finalize': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    'hfl_layer_norm_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                   c_float, c_void_p]),
    'hfl_set_variant': (c_int, [c_char_p, c_int]),
    'hfl_smoothap_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_float, c_void_p]),
    'hfl_gemm_bf16_tn': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'hfl_gemm_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'hfl_window_rpe_expand_size': (c_int64, [c_int, c_int, c_int, c_int]),
    'hfl_window_rpe_expand': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'hfl_relay_attention_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_float, c_int, c_void_p]),
    'hfl_relay_attention_f16_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'hfl_relay_token_init': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                     c_int64, c_void_p]),
    'hfl_window_stats': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p]),
    'hfl_layer_norm': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float,
                               c_void_p]),
    'hfl_add_layer_norm': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_int64, c_float, c_void_p]),
    'hfl_mlp_fused_pack_bytes': (c_int64, [c_int]),
    'hfl_mlp_fused_pack': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    'hfl_ln_mlp_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                         c_void_p]),
    'hfl_ln_mlp_fused_workspace': (c_int64, [c_int64, c_int]),
    'hfl_mlp_fused_pack_bytes_h': (c_int64, [c_int, c_int]),
    'hfl_mlp_fused_pack_h': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'hfl_ln_mlp_fused_workspace_h': (c_int64, [c_int64, c_int, c_int]),
    'hfl_ln_mlp_fused_h': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                   c_int, c_void_p, c_int64, c_void_p]),
    'hfl_ln_mlp_fused_ws': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                            c_void_p, c_int64, c_void_p]),
    'hfl_attn_fused_ok': (c_int, [c_void_p, c_int, c_int]),
    'hfl_attn_fused_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                           c_void_p, c_void_p]),
    'hfl_attn_ws_ok': (c_int, [c_void_p, c_int]),
    'hfl_attn_ws_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_void_p,
                        c_void_p, c_void_p, c_void_p, c_void_p]),
    'hfl_qkv_fused_pack_bytes': (c_int64, [c_int]),
    'hfl_qkv_fused_pack': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'hfl_ln_qkv_fused_seg': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_int64, c_int,
                                     c_void_p]),
    'hfl_linear_x3_seg': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'hfl_ln_qkv_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_float, c_int64, c_int,
                         c_void_p]),
    'hfl_segment_softmax': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
}

_lib = None


def load():
    """Load the shared library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            'libhotformerloc_hip.so not found at %s -- the HIP extension is required '
            '(no CPU fallback); run `python -m hotformerloc_amd.build`' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError('symbol %s missing from %s' % (name, LIB_PATH)) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # probe knobs from the environment (A/B runs of bench.py): HFL_VARIANTS="key=value,key=value" -> hfl_set_variant
    for kv in os.environ.get('HFL_VARIANTS', '').split(','):
        if '=' in kv:
            k, v = kv.split('=', 1)
            if lib.hfl_set_variant(k.strip().encode(), int(v)) != 0:
                raise NativeLibraryError('HFL_VARIANTS: unknown knob %r' % k)
    return lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: 'HFL_EINVAL (unsupported shape/argument)',
                -2: 'HFL_ECAPACITY (input exceeds a kernel limit)'}.get(
                    rc, 'hipBLASLt status %d' % (-100 - rc) if rc <= -100 else 'hipError %d' % rc)
        raise NativeLibraryError('%s failed: %s' % (what, kind))


def ptr_array(ptrs):
    """host array of device pointers (for the pointer-table arguments)"""
    arr = (c_void_p * len(ptrs))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
