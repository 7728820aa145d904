"""CPU-side tests of the host logic around the HIP path (no GPU needed): config parsing,
state_dict layout, window bookkeeping, C-ABI symbol table, loud failure without a GPU."""

import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from hotformerloc_amd import _native, load_config, model_factory, synthetic as syn
from hotformerloc_amd.plan import window_layout
from oracle import hotformer_ref
from oracle.testing import load_case, oracle_octree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name', ['wild-places', 'cs-wild-places', 'oxford', 'cs-campus3d'])
def test_state_dict_matches_reference_layout(golden_dir, name):
    """Names, shapes and order of the reference state_dict; the JSON was
    dumped from the reference model in the build container."""
    params, _ = load_config(name)
    model = model_factory(params)
    spec = json.load(open(os.path.join(golden_dir, 'state_dict_%s.json' % name.replace('-', '_'))))
    mine = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    assert mine == spec
    assert sum(p.numel() for p in model.parameters()) == {
        'wild-places': 35313124, 'cs-wild-places': 35371176, 'oxford': 35329992, 'cs-campus3d': 35415236}[name]


def test_model_params_fields():
    p, depth = load_config('wild-places')
    assert (p.channels, p.num_blocks, p.num_heads) == ((128, 256), (4, 10), (8, 16))
    assert p.patch_size == 48 and p.dilation == 4 and p.ADaPE_mode is None and depth == 7
    assert p.k_pooled_tokens == (148, 72, 36) and p.coordinates == 'cylindrical'
    assert p.normalize_embeddings and p.conv_norm == 'layernorm' and p.drop_path == 0.5
    q, depth = load_config('oxford')
    assert q.ADaPE_mode == 'cov' and q.patch_size == 48 and depth == 9 and q.quantizer is None
    r, _ = load_config('cs-wild-places')
    assert r.patch_size == 64 and r.k_pooled_tokens == (74, 36, 18)


def test_unsupported_options_raise(tmp_path):
    src = open(os.path.join(ROOT, 'hotformerloc_amd', 'configs', 'wild_places.ini')).read()
    bad = tmp_path / 'bad.ini'
    bad.write_text(src.replace('pooling = PyramidAttnPoolMixer', 'pooling = PyramidNetVLAD'))      # as the reference
    from hotformerloc_amd.params import ModelParams
    with pytest.raises(NotImplementedError):
        model_factory(ModelParams(str(bad)))
    assert 'ct_size = 1' in src
    bad.write_text(src.replace('ct_size = 1', 'ct_size = 2'))            # the reference's own model raises on it too
    with pytest.raises(NotImplementedError):
        model_factory(ModelParams(str(bad)))
    bad.write_text(src.replace('ct_propagation = False', 'ct_propagation = True'))       # built since round 3
    assert model_factory(ModelParams(str(bad))) is not None


@pytest.mark.parametrize('case', ['wild_places_ragged', 'cs_wild_places_b2', 'oxford_b2', 'wild_places_b3'])
def test_window_layout_matches_oracle_plan(golden_dir, case):
    """`window_layout` (numpy, product) against the oracle's restatement of OctreeT."""
    g = load_case(golden_dir, case)
    params, depth = load_config(g['cfg'])
    octree = oracle_octree(g['clouds'], depth)
    max_depth = depth - params.num_input_downsamples
    start = max_depth - (params.num_pyramid_levels + params.num_octf_levels) + 1
    plan = hotformer_ref.WindowPlan(octree, params.patch_size, params.dilation, max_depth, start,
                                    params.num_pyramid_levels, params.num_octf_levels,
                                    params.ADaPE_mode)
    lay = window_layout(octree.batch_nnum_nempty.numpy(), params.patch_size, params.dilation,
                        max_depth, start, plan.pyramid_depths)
    B = octree.batch_size
    for d in range(start, max_depth + 1):
        assert lay['n_tokens'][d] == int(plan.nnum_t[d])
        assert lay['n_padded'][d] == int(plan.nnum_a[d])
    for d in plan.pyramid_depths:
        assert np.array_equal(lay['num_windows'][d], plan.batch_num_windows[d].numpy())
        # owner of every window = min batch id; pure padding windows have owner B
        assert lay['n_pad_windows'][d] == int((plan.rt_batch_idx[d] >= B).sum())
    assert np.array_equal(lay['rt_counts'], plan.rt_counts.numpy())
    # the ragged sequences are exactly the unmasked entries of the reference's (B,R,R) mask
    for b in range(B):
        row = plan.rt_attn_mask[b]
        n_real = int((row[0] == 0).sum()) if int(plan.rt_counts[b]) > 0 else 0
        assert lay['seq_off'][b + 1] - lay['seq_off'][b] == n_real
    # each listed row belongs to the cloud that lists it
    owner = np.concatenate([plan.rt_batch_idx[d].numpy() for d in plan.pyramid_depths])
    for b in range(B):
        rows = lay['seq_rows'][lay['seq_off'][b]:lay['seq_off'][b + 1]]
        assert np.all(owner[rows] == b)
    listed = np.zeros(owner.shape[0], bool)
    listed[lay['seq_rows']] = True
    assert np.array_equal(listed, owner < B)


def test_capi_exports_every_declared_symbol():
    """The shared library loads and exports every function include/*.h declares, and the
    ctypes table covers exactly that set (no compute calls: no GPU here)."""
    header = open(os.path.join(ROOT, 'include', 'hotformerloc_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(hfl_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    lib = _native.load()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.hfl_version() == 100
    assert lib.hfl_arch() == b'gfx950'
    assert lib.hfl_octree_scratch_bytes(1000, 2, 600, 7) >= 16 * 1000 * 4
    assert lib.hfl_dwconv_weight_backward_workspace(5000, 256, 27) > 0


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from hotformerloc_amd import Octree, Points, merge_octrees, ops
    o = Octree(7, 2)
    o.build_octree(Points(torch.from_numpy(syn.unit_ball_cloud(1, 256))))     # deferred: fine
    m = merge_octrees([o, o])
    assert m.batch_size == 2 and not m._built
    with pytest.raises(_native.NativeLibraryError):
        m.construct_all_neigh()
    with pytest.raises(_native.NativeLibraryError):
        ops.dwconv_forward_backward(torch.zeros(4, 8), torch.zeros(27, 1, 8),
                                    torch.zeros(4, 27, dtype=torch.int64))
    params, _ = load_config('wild-places')
    with pytest.raises(RuntimeError):
        model_factory(params)({'octree': m})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'hotformerloc_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
    for f in ('bench.py', '__graft_entry__.py'):
        path = os.path.join(ROOT, f)
        if os.path.exists(path):
            assert '/root/reference' not in open(path).read()


def test_synthetic_generators_are_pinned():
    u = syn.hash_uniform(12345, 4)
    assert np.allclose(u, syn.hash_uniform(12345, 6)[:4]) and np.all(np.abs(u) < 1)
    pc = syn.unit_ball_cloud(1000, 4096)
    assert pc.shape == (4096, 3) and pc.dtype == np.float32
    assert np.all(np.linalg.norm(pc, axis=1) < 1.0)
    assert abs(float(pc.astype(np.float64).sum()) - float(syn.unit_ball_cloud(1000, 4096).astype(np.float64).sum())) == 0
    cyl = syn.cylindrical(pc)
    assert np.all(np.abs(cyl) <= 1.0)
    w = syn.synthetic_tensor('backbone.backbone.octf_stage.0.blocks.0.attention.qkv.weight', (384, 128))
    assert w.shape == (384, 128) and 0.05 < w.std() < 0.11
    assert syn.synthetic_tensor('x.norm1.weight', (128,)).mean() > 0.9
    f = syn.forest_cloud(5, 6000)
    assert f.shape == (6000, 3) and np.all(np.abs(f) < 1)


def test_split_ranges_matches_the_python_loop():
    """Row tiles / pair chunks of the tap kernels (octree._split_ranges): consecutive ranges cut into pieces of <= step, in
    order, never across a range boundary -- against the obvious double loop, with empty ranges and exact multiples."""
    from hotformerloc_amd.octree import _split_ranges
    rng = np.random.default_rng(3)
    for step in (128, 2048):
        for _ in range(20):
            cnt = rng.integers(0, 5 * step, size=rng.integers(1, 30))
            cnt[rng.integers(0, cnt.size)] = 0
            cnt[rng.integers(0, cnt.size)] = 2 * step
            e = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
            rid, first, length = _split_ranges(e, step)
            want = [(k, a, min(step, e[k + 1] - a)) for k in range(cnt.size) for a in range(e[k], e[k + 1], step)]
            assert [(int(a), int(b), int(c)) for a, b, c in zip(rid, first, length)] == [tuple(int(v) for v in w) for w in want]


def test_split_weight_cache_releases_with_the_parameter():
    """The split2 / split3 copies of a Linear weight are as large as the weight: they must go when the parameter goes
    (the 'x3' entries were never evicted)."""
    import gc
    import torch
    from hotformerloc_amd import model as M
    before = len(M._W3_CACHE)
    lin = torch.nn.Linear(64, 32)
    w2 = M._w2(lin)
    assert w2.shape == (32, 128) and M._w2(lin) is w2            # cached
    w3 = M._w3(lin)
    assert w3.shape == (32, 192) and len(M._W3_CACHE) == before + 2
    with torch.no_grad():
        lin.weight.add_(1.0)                                      # in-place update: version bump -> re-split, no new entry
    assert M._w2(lin) is not w2 and len(M._W3_CACHE) == before + 2
    del lin, w2, w3
    gc.collect()
    assert len(M._W3_CACHE) == before


def test_tall_mm_split_k_weight_gradient_equals_autograd():
    """autograd.TallMmFn (the dense-gather convolutions' col @ w with a split-K weight gradient, ocnn's octree2col + mm
    of models/layers/octformer_layers.py:89-95): outputs and all three gradients equal plain autograd, for row counts below one
    chunk, whole chunks, and chunks plus a remainder; with and without bias."""
    import torch
    from hotformerloc_amd.autograd import TallMmFn, tall_mm
    old, TallMmFn.CHUNK = TallMmFn.CHUNK, 64
    try:
        g = torch.Generator().manual_seed(0)
        for n in (10, 128, 1000):
            col = torch.randn(n, 81, generator=g, requires_grad=True, dtype=torch.float64)
            w = torch.randn(81, 32, generator=g, requires_grad=True, dtype=torch.float64)
            b = torch.randn(32, generator=g, requires_grad=True, dtype=torch.float64)
            dy = torch.randn(n, 32, generator=g, dtype=torch.float64)
            for bias in (None, b):
                y = tall_mm(col, w, bias)
                y.backward(dy)
                got = (col.grad.clone(), w.grad.clone(), None if bias is None else b.grad.clone())
                col.grad = w.grad = b.grad = None
                y2 = (col @ w) + (0 if bias is None else bias)
                y2.backward(dy)
                assert torch.allclose(y, y2) and torch.allclose(got[0], col.grad) and torch.allclose(got[1], w.grad)
                if bias is not None:
                    assert torch.allclose(got[2], b.grad)
                col.grad = w.grad = b.grad = None
    finally:
        TallMmFn.CHUNK = old


def test_ctypes_structures_match_the_c_header(tmp_path):
    """Every struct the C-ABI passes by pointer (include/hotformerloc_hip.h) against its ctypes mirror in _native.py: a C program
    compiled here with gcc prints sizeof and every member's offset; both must agree field by field (a silent mismatch would
    hand the library shifted pointers)."""
    import ctypes
    import shutil
    import subprocess
    from hotformerloc_amd import _native
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    pairs = [('hfl_window_attn_desc', _native.WindowAttnDesc), ('hfl_row_segments', _native.RowSegments),
             ('hfl_block_weights', _native.BlockWeights), ('hfl_block_io', _native.BlockIO),
             ('hfl_relay_block_weights', _native.RelayBlockWeights), ('hfl_relay_block_io', _native.RelayBlockIO)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hotformerloc_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include')
    subprocess.run(['gcc', '-std=c99', '-I', inc, str(src), '-o', str(exe)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        a, b, c = ln.split()
        got[a, b] = int(c)
    for cname, cls in pairs:
        assert got[cname, 'sizeof'] == ctypes.sizeof(cls), (cname, got[cname, 'sizeof'], ctypes.sizeof(cls))
        n_c = sum(1 for k in got if k[0] == cname) - 1
        assert n_c == len(cls._fields_)
        for fname, _ in cls._fields_:
            assert got[cname, fname] == getattr(cls, fname).offset, (cname, fname)
