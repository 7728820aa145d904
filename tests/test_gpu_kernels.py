"""GPU parity tests, kernel by kernel: HIP path (through the C-ABI) vs the CPU oracle on the
same seeded inputs.  Bit-exact for integer/index work, stated tolerances for fp32."""

import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hotformerloc_amd import _native, Octree, Points, merge_octrees, build_batch_octree, load_config, ops
from hotformerloc_amd import dwconv as hdw
from hotformerloc_amd import synthetic as syn
from hotformerloc_amd.plan import WindowPlan
from oracle import hotformer_ref
from oracle.ocnn_ref import nn as onn
from oracle.ocnn_ref.octree import Octree as OOctree, Points as OPoints, merge_octrees as omerge
from oracle.testing import oracle_octree

DEV = 'cuda'


def _fixture_clouds(golden_dir, ids=(4, 5)):
    out = []
    for i in ids:
        d = np.load(os.path.join(golden_dir, 'ocnn', 'test_%03d.npz' % i))
        out.append(d['points'])
    return out


# ------------------------------------------------------------------------- octree
def _compare_octree(dev_oct, ref_oct, check_points=True):
    D = ref_oct.depth
    assert np.array_equal(dev_oct.nnum.numpy(), ref_oct.nnum.numpy())
    assert np.array_equal(dev_oct.nnum_nempty.numpy(), ref_oct.nnum_nempty.numpy())
    assert np.array_equal(dev_oct.batch_nnum_nempty.numpy(), ref_oct.batch_nnum_nempty.numpy())
    assert np.array_equal(dev_oct.batch_nnum.numpy(), ref_oct.batch_nnum.numpy())
    for d in range(D + 1):
        assert torch.equal(dev_oct.keys[d].cpu(), ref_oct.keys[d]), 'keys depth %d' % d
        assert torch.equal(dev_oct.children[d].cpu(), ref_oct.children[d]), 'children depth %d' % d
        assert torch.equal(dev_oct.key(d, nempty=True).cpu(), ref_oct.key(d, nempty=True))
    for d in range(1, D + 1):
        want = ref_oct.get_neigh(d, '333', 1, nempty=True)
        got = dev_oct.get_neigh(d, '333', 1, nempty=True).cpu().long()
        assert torch.equal(got, want), 'neigh depth %d' % d
    for d in range(3, D + 1):
        assert torch.equal(dev_oct.get_neigh(d, '222', 2, nempty=True).cpu().long(),
                           ref_oct.get_neigh(d, '222', 2, nempty=True))
    if check_points:
        a = dev_oct.get_input_feature('P', True).cpu()
        b = ref_oct.get_input_feature('P', nempty=True)
        assert (a - b).abs().max() < 1e-6


def test_octree_build_matches_ocnn_fixture(golden_dir):
    """Batch of ocnn's test clouds 4+5 (depth 6, full depth 3): keys/children/counts bit-exact
    against the vendored ocnn golden file batch_45.npz, neighbours against the oracle."""
    clouds = _fixture_clouds(golden_dir)
    o = build_batch_octree(clouds, depth=6, full_depth=3, device=DEV)
    b = np.load(os.path.join(golden_dir, 'ocnn', 'batch_45.npz'))
    assert np.array_equal(torch.cat(o.keys).cpu().numpy(), b['key'])
    assert np.array_equal(torch.cat(o.children).cpu().numpy(), b['child'])
    assert np.array_equal(o.nnum.numpy(), b['nnum'])
    assert np.array_equal(o.nnum_nempty.numpy(), b['nnum_nempty'])
    ref = omerge([_obuild(c, 6, 3) for c in clouds])
    ref.construct_all_neigh()
    _compare_octree(o, ref, check_points=False)
    # fixture neighbour table (all nodes): rows of non-empty nodes, remapped, equal ours
    neigh_all = torch.from_numpy(b['neigh']).long()
    row0 = 0
    for d in range(1, 7):
        n = int(b['nnum'][d])
        tab = neigh_all[row0:row0 + n]
        row0 += n
        child = ref.children[d]
        tab = tab[child >= 0]
        valid = tab >= 0
        tab[valid] = child[tab[valid]].long()
        assert torch.equal(o.get_neigh(d, '333', 1, True).cpu().long(), tab)


def _obuild(pc, depth, full_depth=2):
    o = OOctree(depth, full_depth)
    o.build_octree(OPoints(torch.from_numpy(np.ascontiguousarray(pc, dtype=np.float32))))
    return o


@pytest.mark.parametrize('depth,sizes,kind', [
    (7, [4096, 4096, 4096], 'ball'), (9, [4096, 3000], 'ball'), (7, [9, 60, 1, 700, 16384], 'mixed'),
    (5, [5], 'ball'), (10, [8192], 'forest')])
def test_octree_build_matches_oracle(depth, sizes, kind):
    clouds = []
    for i, n in enumerate(sizes):
        seed = 40000 + 100 * depth + i
        pc = syn.forest_cloud(seed, n) if (kind == 'forest' or (kind == 'mixed' and i % 2)) \
            else syn.unit_ball_cloud(seed, n)
        clouds.append(pc)
    ref = oracle_octree(clouds, depth)
    dev = build_batch_octree(clouds, depth, 2, DEV)
    _compare_octree(dev, ref)


def test_octree_api_deferred_build_and_merge():
    """The reference's call sequence (dataset_utils.py:89-94 + torch_utils.py:47-51)."""
    clouds = [syn.unit_ball_cloud(7 + i, 1000 + 300 * i) for i in range(3)]
    octs = []
    for pc in clouds:
        o = Octree(7, 2)
        o.build_octree(Points(torch.from_numpy(pc)))       # CPU points: deferred
        octs.append(o)
    m = merge_octrees(octs)
    assert m.batch_size == 3
    m = m.to(DEV)
    m.construct_all_neigh()
    _compare_octree(m, oracle_octree(clouds, 7))
    back = m.cpu()
    assert back.children[7].device.type == 'cpu' and torch.equal(back.keys[5], m.keys[5].cpu())
    # coordinates exactly on the upper bound wrap like ocnn's masked key (documented edge case)
    edge = np.array([[0.999, -1.0, 0.5], [1.0, 0.25, -0.25]], dtype=np.float32)
    e = build_batch_octree([edge], 6, 2, DEV)
    r = oracle_octree([edge], 6)
    assert torch.equal(e.key(6, True).cpu(), r.key(6, True))


def test_octree_large_cloud_multi_chunk_sort():
    """> 16384 points: the bitonic network's wide strides run through L2 (two and four chunks)."""
    clouds = [syn.unit_ball_cloud(1, 40000), syn.forest_cloud(2, 16385), syn.unit_ball_cloud(3, 100)]
    _compare_octree(build_batch_octree(clouds, 8, 2, DEV), oracle_octree(clouds, 8))


@pytest.mark.parametrize('rows,taps', [(1, 27), (1023, 27), (1024, 8), (70001, 27), (5000, 3)])
def test_tap_lists_match_tap_major_compaction(rows, taps):
    """hfl_tap_lists (the live (row, tap) pairs of an octree-conv table, tap-major, rows ascending inside a tap)
    against the plain definition; integer work -> bit-exact.  Includes an all-dead tap and an all-live tap."""
    g = torch.Generator().manual_seed(rows * 31 + taps)
    table = torch.randint(0, max(rows, 2), (rows, taps), generator=g, dtype=torch.int32)
    dead = torch.rand((rows, taps), generator=g) < 0.7
    dead[:, 0] = True                      # tap 0 has no neighbour anywhere
    dead[:, taps - 1] = False              # the last tap is live everywhere
    table[dead] = -1
    src, slot, edges = ops.tap_lists(table.to(DEV))
    edges = edges.cpu()
    live_t = (table >= 0).t()
    counts = live_t.sum(1)
    want_edges = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(counts, 0)]).to(torch.int32)
    assert torch.equal(edges, want_edges)
    want_src = table.t()[live_t]
    assert torch.equal(src[:int(edges[-1])].cpu(), want_src)
    rank = torch.cumsum(live_t.reshape(-1).to(torch.int64), 0) - 1
    want_slot = torch.where(live_t, rank.view(taps, rows), torch.full((taps, rows), -1)).t().to(torch.int32)
    assert torch.equal(slot.cpu(), want_slot)


def test_tap_lists_multi_equals_single_table_calls():
    """hfl_tap_lists_multi: the tables of every convolution depth of a batch in three launches in all -- the same lists as
    one hfl_tap_lists call per table (27-tap, 8-tap and odd-width tables, one of them a single row), bit-exact."""
    g = torch.Generator().manual_seed(11)
    tables = []
    for rows, taps in ((70001, 27), (5000, 8), (1, 27), (2049, 8), (3000, 5), (130000, 27)):
        t = torch.randint(0, max(rows, 2), (rows, taps), generator=g, dtype=torch.int32)
        t[torch.rand((rows, taps), generator=g) < 0.75] = -1
        tables.append(t.to(DEV))
    edges = [torch.empty(t.shape[1] + 1, dtype=torch.int32, device=DEV) for t in tables]
    multi = ops.tap_lists_multi(tables, edges)
    for t, (src, slot, e) in zip(tables, multi):
        src1, slot1, e1 = ops.tap_lists(t)
        assert torch.equal(e, e1) and torch.equal(slot, slot1)
        n = int(e1[-1])
        assert torch.equal(src[:n], src1[:n])


def test_octree_sparse_taps_built_with_the_neighbour_tables():
    """construct_all_neigh() launches the live-tap lists of all octree convolutions and starts ONE asynchronous device->host
    read of their counts; the first sparse_taps() collects it for every table at once, nothing else is built later."""
    clouds = syn.make_clouds(9, 3, 3000, 'cartesian')
    o = build_batch_octree(clouds, 7, 2, DEV, construct_neigh=True)
    keys = [(6, '333', 1), (5, '333', 1), (7, '222', 2), (6, '222', 2), (5, '222', 2), (4, '222', 2), (3, '222', 2)]
    assert set(o.__dict__['_taps_pending'][0]) >= set(keys)
    o.sparse_taps(*keys[0])
    assert '_taps_pending' not in o.__dict__
    cache = o.__dict__['_sparse_taps']
    for key in keys:
        assert key in cache, key
        src, slot, edges = cache[key]
        tiles = o.tap_tiles(*key, 128).cpu().numpy()              # device-built row tiles of the grouped tap GEMM
        e = np.asarray(edges)
        want = [(a, min(128, e[k + 1] - a), k * 128) for k in range(len(e) - 1) for a in range(e[k], e[k + 1], 128)]
        assert tiles.shape == (len(want), 3) and (tiles == np.asarray(want, dtype=np.int32).reshape(-1, 3)).all(), key
        table = o.get_neigh(key[0], key[1], key[2], nempty=True)
        live = table >= 0
        assert edges[-1] == int(live.sum()) == src.shape[0] and len(edges) == table.shape[1] + 1
        # every live pair points at its own source row
        r, k = torch.nonzero(live, as_tuple=True)
        assert torch.equal(src[slot[r, k].long(), 0], table[r, k])
        assert bool((slot[~live] == -1).all())


def test_live_tap_conv_gradients_match_the_dense_formulation():
    """Training path of OctreeConv over its live taps (autograd.LiveTapConvFn) against autograd over the dense
    octree2col + GEMM formulation of the same module: output, input gradient and weight gradient, 3x3x3 stride 1 and
    2x2x2 stride 2 (children table: source and destination depths differ)."""
    from hotformerloc_amd import model as M
    clouds = syn.make_clouds(13, 3, 3000, 'cartesian')
    o = build_batch_octree(clouds, 7, 2, DEV, construct_neigh=True)
    g = torch.Generator().manual_seed(3)
    for depth, ks, stride, cin, cout in ((6, [3], 1, 64, 64), (5, [3], 1, 128, 128), (6, [2], 2, 64, 128)):
        conv = M.OctreeConv(cin, cout, ks, stride, nempty=True, use_bias=stride == 2).to(DEV)
        n_in = int(o.nnum_nempty[depth])
        x = torch.randn(n_in, cin, generator=g).to(DEV)
        got = {}
        for mode in (True, False):
            M._SPARSE_CONV = mode
            try:
                xi = x.clone().requires_grad_(True)
                conv.zero_grad()
                y = conv(xi, o, depth)
                dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(4)).to(DEV)
                y.backward(dy)
                got[mode] = (y.detach(), xi.grad.clone(), conv.weights.grad.clone())
            finally:
                M._SPARSE_CONV = True
        for a, b, what in zip(got[True], got[False], ('out', 'ddata', 'dweights')):
            assert a.shape == b.shape
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-6, (depth, ks, stride, what)


def test_slot_sum_equals_the_unit_weight_convolution():
    """hfl_slot_sum (the per-row sum of a live-tap convolution's partial products, bias included) is bitwise the depth-wise
    convolution with unit weights it replaced, plus the bias."""
    g = torch.Generator().manual_seed(12)
    for n, p, c, k in ((1, 5, 256, 27), (1000, 7000, 64, 27), (4099, 9000, 128, 8), (300, 900, 32, 27)):
        slot = torch.randint(-1, p, (n, k), generator=g, dtype=torch.int32).to(DEV)
        part = torch.randn(p, c, generator=g).to(DEV)
        bias = torch.randn(c, generator=g).to(DEV)
        want = ops.dwconv_forward_backward(part, torch.ones(k, 1, c, device=DEV), slot)
        assert torch.equal(ops.slot_sum(part, slot), want), (n, c, k)
        assert torch.equal(ops.slot_sum(part, slot, bias), want + bias), (n, c, k)


def test_dwconv_weight_gradient_at_small_and_ragged_row_counts():
    """hfl_dwconv_weight_backward deals the rows to the eight XCDs in contiguous eighths: row counts below one workgroup per
    XCD, one row, and counts that are no multiple of the rows per workgroup must still visit every row once."""
    g = torch.Generator().manual_seed(11)
    for n, c, k in ((1, 256, 27), (7, 256, 27), (50, 128, 27), (131, 64, 8), (1000, 256, 27), (4099, 32, 27)):
        neigh = torch.randint(-1, n, (n, k), generator=g, dtype=torch.int32)
        x, dy = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
        ref = torch.zeros(k, c, dtype=torch.float64)
        for t in range(k):
            ok = neigh[:, t] >= 0
            ref[t] = (x[neigh[ok, t].long()].double() * dy[ok].double()).sum(0)
        got = ops.dwconv_weight_backward(dy.to(DEV), x.to(DEV), neigh.to(DEV)).view(k, c).cpu().double()
        assert (got - ref).abs().max().item() <= 1e-5 * max(ref.abs().max().item(), 1.0), (n, c, k)


def test_tap_wgrad_through_row_tables_equals_the_pair_major_form():
    """hfl_tap_wgrad_gather: the weight gradient read through the pair tables (layer input by `src`, output gradient by
    `rowof`) is BITWISE the one computed from the two pair-major copies (same kernel, same summation order), and both match
    the fp64 contraction."""
    clouds = syn.make_clouds(21, 3, 3000, 'cartesian')
    o = build_batch_octree(clouds, 7, 2, DEV, construct_neigh=True)
    g = torch.Generator().manual_seed(5)
    for depth, kernel, stride, cin, cout in ((6, '333', 1, 64, 64), (5, '333', 1, 128, 64), (6, '222', 2, 64, 128)):
        src, slot, edges = o.sparse_taps(depth, kernel, stride)
        rowof, _, chunks, tap_off = o.sparse_taps_bwd(depth, kernel, stride)
        x = torch.randn(int(o.nnum_nempty[depth]), cin, generator=g).to(DEV)
        dy = torch.randn(slot.shape[0], cout, generator=g).to(DEV)
        gp, dp = ops.octree_gather(x, src), ops.octree_gather(dy, rowof)
        taps = slot.shape[1]
        a = ops.tap_wgrad(gp, dp, chunks, tap_off, taps)
        b = ops.tap_wgrad(x, dy, chunks, tap_off, taps, g_rows=src, d_rows=rowof)
        c = ops.tap_wgrad(x, dp, chunks, tap_off, taps, g_rows=src)
        assert torch.equal(a, b) and torch.equal(a, c), (depth, kernel)
        ref = torch.stack([gp[edges[k]:edges[k + 1]].double().t() @ dp[edges[k]:edges[k + 1]].double() for k in range(taps)])
        assert ((a.double() - ref).norm() / ref.norm()).item() < 1e-6


# ------------------------------------------------------------------------- dwconv
def test_dwconv_matches_ocnn_semantics(golden_dir):
    """Port of the reference's own test (`libs/dwconv/test/test_octree_dwconv.py:13-68`):
    depth 4, C=256, 8 kernel shapes x nempty in {True, False}; the three raw functions, the
    autograd Function and the Module against `ocnn.nn.OctreeDWConv` (oracle) fwd + both grads;
    tolerances from the reference test: 1e-6 (out, data grad), 5e-5 (weight grad)."""
    depth, channel = 4, 256
    ref = omerge([_obuild(c, 6, 3) for c in _fixture_clouds(golden_dir)])
    ref.construct_all_neigh()
    kernel_size = [[3, 3, 3], [3, 1, 1], [1, 3, 1], [1, 1, 3], [2, 2, 2], [3, 3, 1], [1, 3, 3], [3, 1, 3]]
    g = torch.Generator().manual_seed(0)
    for ks in kernel_size:
        for nempty in (True, False):
            nnum = int(ref.nnum_nempty[depth] if nempty else ref.nnum[depth])
            rnd = torch.randn(nnum, channel, generator=g)
            oc = onn.OctreeDWConv(channel, ks, nempty=nempty)
            od = rnd.clone().requires_grad_()
            oout = oc(od, ref, depth)
            oout.sum().backward()
            kernel = ''.join(str(k) for k in ks)
            neigh = ref.get_neigh(depth, kernel, 1, nempty).to(DEV)          # int64, as the reference
            data = rnd.clone().to(DEV).requires_grad_()
            weights = oc.weights.detach().clone().to(DEV).requires_grad_()
            out = hdw.dwconv_forward_backward(data.detach(), weights.detach(), neigh)
            grad = torch.ones_like(out)
            ineigh = hdw.inverse_neigh(neigh)
            grad_d = hdw.dwconv_forward_backward(grad, weights.detach(), ineigh)
            grad_w = hdw.dwconv_weight_backward(grad, data.detach(), neigh)
            assert torch.allclose(out.cpu(), oout.detach(), atol=1e-6), (ks, nempty)
            assert torch.allclose(grad_d.cpu(), od.grad, atol=1e-6), (ks, nempty)
            assert torch.allclose(grad_w.cpu(), oc.weights.grad, atol=5e-5), (ks, nempty)
            o2 = hdw.octree_dwconv(data, weights, neigh)
            o2.sum().backward()
            assert torch.allclose(o2.detach().cpu(), oout.detach(), atol=1e-6)
            assert torch.allclose(data.grad.cpu(), od.grad, atol=1e-6)
            assert torch.allclose(weights.grad.cpu(), oc.weights.grad, atol=5e-5)
            # int32 tables (what the library builds itself) give identical results
            o3 = hdw.dwconv_forward_backward(data.detach(), weights.detach(), neigh.int())
            assert torch.equal(o3, out)
            assert torch.equal(hdw.inverse_neigh(neigh.int()).long(), ineigh)


def test_dwconv_module_and_odd_channels():
    clouds = [syn.unit_ball_cloud(91, 2000), syn.forest_cloud(92, 1500)]
    ref = oracle_octree(clouds, 6)
    dev = build_batch_octree(clouds, 6, 2, DEV)
    g = torch.Generator().manual_seed(1)
    for C in (96, 30, 7):       # vector path with 24 lanes/row, 4-aligned small, scalar path
        n = int(ref.nnum_nempty[5])
        x = torch.randn(n, C, generator=g)
        mod = hdw.OctreeDWConv(C, [3], nempty=True).to(DEV)
        oc = onn.OctreeDWConv(C, [3], nempty=True)
        with torch.no_grad():
            oc.weights.copy_(mod.weights.cpu())
        xo = x.clone().requires_grad_()
        oo = oc(xo, ref, 5)
        (oo * oo).sum().backward()
        xd = x.clone().to(DEV).requires_grad_()
        od = mod(xd, dev, 5)
        (od * od).sum().backward()
        assert torch.allclose(od.detach().cpu(), oo.detach(), atol=2e-6, rtol=1e-5)
        assert torch.allclose(xd.grad.cpu(), xo.grad, atol=1e-5, rtol=1e-4)
        assert torch.allclose(mod.weights.grad.cpu(), oc.weights.grad, atol=2e-3, rtol=1e-4)


def test_cpe_fused_and_gather():
    clouds = [syn.unit_ball_cloud(11, 3000), syn.unit_ball_cloud(12, 2500)]
    ref = oracle_octree(clouds, 7)
    dev = build_batch_octree(clouds, 7, 2, DEV)
    g = torch.Generator().manual_seed(2)
    for depth, C in ((5, 128), (4, 256), (3, 256)):
        n = int(ref.nnum_nempty[depth])
        x = torch.randn(n, C, generator=g)
        sd = {'c.conv.weights': torch.randn(27, 1, C, generator=g) * 0.3,
              'c.norm.weight': 1 + 0.1 * torch.randn(C, generator=g),
              'c.norm.bias': 0.1 * torch.randn(C, generator=g)}
        want = hotformer_ref.cpe(x, sd, 'c', ref, depth)
        neigh = dev.get_neigh(depth, '333', 1, True)
        lib = _native.load()
        try:
            for variant in (0, 1):                     # direct gathers / LDS-staged de-duplicated gathers
                lib.hfl_set_variant(b'cpe_variant', variant)
                for residual in (False, True):
                    got = ops.cpe_forward(x.to(DEV), sd['c.conv.weights'].to(DEV), sd['c.norm.weight'].to(DEV),
                                          sd['c.norm.bias'].to(DEV), neigh, residual).cpu()
                    w = want + x if residual else want
                    assert torch.allclose(got, w, atol=2e-5, rtol=1e-5), (depth, C, residual, variant)
        finally:
            lib.hfl_set_variant(b'cpe_variant', 0)
    # gather == octree2col, both neighbour kinds and a 3-channel input
    for depth, C, kernel, stride in ((7, 3, '333', 1), (6, 64, '333', 1), (6, 64, '222', 2), (7, 32, '222', 2)):
        n = int(ref.nnum_nempty[depth])
        x = torch.randn(n, C, generator=g)
        want = onn.octree_gather(x, ref.get_neigh(depth, kernel, stride, True)).flatten(1)
        got = ops.octree_gather(x.to(DEV), dev.get_neigh(depth, kernel, stride, True).contiguous()).cpu()
        assert torch.equal(got, want)


def test_cpe_training_function_matches_fp64_autograd_of_the_reference_formula():
    """autograd.CpeFn (the fused CPE launch as the training forward, keeping the convolution's output; backward = LayerNorm
    backward + dwconv data / weight gradients + the skip gradient) against fp64 torch autograd through the reference's
    formula  [x +] LayerNorm(sum_k w[k] * x[neigh[:, k]])  (models/layers/octformer_layers.py:138-142,
    libs/dwconv/dwconv/nn.py:17-43), and against the three-launch training path it replaces."""
    from hotformerloc_amd import autograd as ag
    from hotformerloc_amd import dwconv as hdw
    clouds = [syn.unit_ball_cloud(31, 3000), syn.unit_ball_cloud(32, 1800)]
    dev = build_batch_octree(clouds, 7, 2, DEV)
    g = torch.Generator(device='cuda').manual_seed(4)
    for depth, C in ((5, 128), (4, 256), (3, 256)):
        neigh = dev.get_neigh(depth, '333', 1, True).contiguous()
        n = neigh.shape[0]
        for residual in (True, False):
            leaves = [torch.randn(n, C, device='cuda', generator=g), torch.randn(27, 1, C, device='cuda', generator=g) * 0.3,
                      1 + 0.1 * torch.randn(C, device='cuda', generator=g), 0.1 * torch.randn(C, device='cuda', generator=g)]
            proj = torch.randn(n, C, device='cuda', generator=g)

            def run(kind):
                x, w, gm, bt = [t.detach().clone().double().requires_grad_() if kind == 'ref' else t.detach().clone().requires_grad_()
                                for t in leaves]
                if kind == 'fused':
                    y = ag.cpe(x, w, gm, bt, neigh, residual, 1e-5)
                elif kind == 'three':
                    y = ag.layer_norm(hdw.octree_dwconv(x, w, neigh), gm, bt, 1e-5)
                    y = x + y if residual else y
                else:
                    idx = neigh.long()
                    col = torch.cat([x, x.new_zeros(1, C)], 0)[torch.where(idx >= 0, idx, n)]      # (n, 27, C)
                    conv = (col * w.view(1, 27, C)).sum(1)
                    y = torch.nn.functional.layer_norm(conv, (C,), gm, bt, 1e-5)
                    y = x + y if residual else y
                (y * (proj.double() if kind == 'ref' else proj)).sum().backward()
                return [y.detach()] + [t.grad for t in (x, w, gm, bt)]
            ref, fused, three = run('ref'), run('fused'), run('three')
            for name, r, f, t in zip(('out', 'dx', 'dw', 'dgamma', 'dbeta'), ref, fused, three):
                scale = r.abs().max().item()
                ef, et = (f.double() - r).abs().max().item() / scale, (t.double() - r).abs().max().item() / scale
                assert ef < 2e-5 and et < 2e-5, (depth, C, residual, name, ef, et)


# ---------------------------------------------------------------------- attention
def _plans(clouds, cfg, octree_depth):
    params, _ = load_config(cfg)
    ref = oracle_octree(clouds, octree_depth)
    dev = build_batch_octree(clouds, octree_depth, 2, DEV)
    md = octree_depth - 2
    args = dict(patch_size=params.patch_size, dilation=params.dilation, max_depth=md,
                start_depth=md - 3, num_pyramid_levels=3, num_octf_levels=1,
                adape_mode=params.ADaPE_mode)
    return params, ref, dev, hotformer_ref.WindowPlan(ref, **args), WindowPlan(dev, **args)


def _pack_qkv_f16(qkv: torch.Tensor, H: int, q_scale: float) -> torch.Tensor:
    """fp32 (rows, 3C) [q | k | v] -> the operand layout of hfl_linear_x3_qkv (csrc/gemm_x3.hip EPI 2): per region and
    head [16 x hi | 16 x lo] fp16 (any split with hi + lo = v to 22 bits is valid), q times q_scale; returned as an
    opaque float32 (rows, 3C) buffer."""
    rows, c3 = qkv.shape
    C = c3 // 3
    x = qkv.clone().float()
    x[:, :C] *= q_scale
    x = x.view(rows, 3, H, 16)
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    packed = torch.stack([hi, lo], dim=3).contiguous()            # (rows, 3, H, 2, 16) fp16 = 12 C bytes per row
    return packed.view(rows, -1).view(torch.float32).view(rows, c3).contiguous()


def test_linear_x3_qkv_epilogue_writes_the_attention_operand_layout():
    """hfl_linear_x3_qkv == split(hfl_linear_x3 output) in the layout `_pack_qkv_f16` describes: hi + lo reproduces the
    fp32 projection (queries scaled) to fp16-pair precision."""
    g = torch.Generator().manual_seed(31)
    for n, C, H in ((777, 128, 8), (1300, 256, 16)):
        x = torch.randn(n, C, generator=g)
        w = torch.randn(3 * C, C, generator=g) * 0.08
        b = torch.randn(3 * C, generator=g) * 0.1
        x2, w2 = ops.split2(x.to(DEV)), ops.split2_weight(w.to(DEV))
        ref = ops.linear_x3(x2, w2, bias=b.to(DEV)).cpu()
        ref[:, :C] *= 0.36
        got = ops.linear_x3_qkv(x2, w2, b.to(DEV), 0.36).cpu()
        halves = got.view(torch.float16).view(n, 3, H, 2, 16).float()
        rec = (halves[:, :, :, 0] + halves[:, :, :, 1]).reshape(n, 3 * C)
        assert (rec - ref).abs().max().item() <= 2 ** -20 * ref.abs().max().item() + 1e-7
        assert (halves[:, :, :, 1].abs() <= halves[:, :, :, 0].abs() * 2 ** -9 + 1e-6).all()      # lo is the residual


@pytest.mark.parametrize('cfg,sizes,odepth', [('wild-places', [4096, 30, 2500], 7), ('cs-wild-places', [5000, 3000], 7),
                                              ('oxford', [4096, 2500], 9)])
def test_window_attention_matches_oracle(cfg, sizes, odepth):
    """odepth 9 (Oxford): attention at octree depths 7..4 -- coordinates beyond pos_bnd (the reference's clamp is live)
    and, on the fp16 path, the three-table form of the expanded RPE table."""
    clouds = [syn.unit_ball_cloud(500 + i, n) for i, n in enumerate(sizes)]
    params, ref, dev, oplan, plan = _plans(clouds, cfg, odepth)
    K, D = params.patch_size, params.dilation
    g = torch.Generator().manual_seed(3)
    B = len(sizes)
    md = odepth - 2
    for depth, H, G, dil in ((md, 8, 0, 1), (md, 8, 0, D), (md - 1, 16, 1, 1), (md - 2, 16, 1, 1), (md - 3, 16, 1, 1)):
        C = H * 16
        nt, W = plan.n_tokens[depth], plan.n_windows[depth]
        assert nt == int(oplan.nnum_t[depth]) and W == int(oplan.nnum_a[depth]) // K
        qkv_tok = torch.randn(nt, 3 * C, generator=g)
        qkv_rt = torch.randn(W, 3 * C, generator=g)
        bnd = int(0.8 * K * dil ** 0.5)
        table = torch.randn(3 * (2 * bnd + 1), H, generator=g) * 0.5
        # oracle on the reference's materialised windows
        xw = oplan.to_windows(qkv_tok, depth, dil > 1)
        if G:
            xw = torch.cat([qkv_rt.unsqueeze(1), xw], 1)
        q, k, v = xw.reshape(-1, K + G, 3, H, 16).permute(2, 0, 3, 1, 4)
        if dil > 1:
            mask, pos = oplan.dilate_mask[depth], oplan.dilate_pos[depth]
        else:
            mask, pos = (oplan.hat_mask[depth] if G else oplan.patch_mask[depth]), oplan.rel_pos[depth]
        rpe = hotformer_ref.rpe_bias(table, pos, K, dil)
        if G:
            rpe = torch.nn.functional.pad(rpe, (G, 0, G, 0))
        want = hotformer_ref._sdpa(q, k, v, mask.unsqueeze(1) + rpe, 0.25).transpose(1, 2).reshape(-1, K + G, C)
        want_tok = oplan.from_windows(want[:, G:], depth, dil > 1)
        got = ops.window_attention(torch.cat([qkv_tok, qkv_rt]).to(DEV), plan.meta[depth],
                                   table.to(DEV), nt, W, K, dil, G, H, B, rt_row0=nt).cpu()
        err = (got[:nt] - want_tok).abs().max().item()
        assert err < 2e-5, (cfg, depth, G, dil, err)
        # depth given: the kernel may drop the RPE clamp (coordinates < 2^depth <= pos_bnd)
        got_d = ops.window_attention(torch.cat([qkv_tok, qkv_rt]).to(DEV), plan.meta[depth],
                                     table.to(DEV), nt, W, K, dil, G, H, B, rt_row0=nt, depth=depth).cpu()
        err = (got_d[:nt] - want_tok).abs().max().item()      # two-lookup kernel (expanded table)
        assert err < 2e-5, ('depth given', cfg, depth, G, dil, err)
        if G:
            real_w = -(-nt // K)
            assert (got_d[nt:nt + real_w] - want[:real_w, 0]).abs().max().item() < 2e-5
        got_s = ops.window_attention(torch.cat([qkv_tok, qkv_rt]).to(DEV), plan.meta[depth],
                                     table.to(DEV), nt, W, K, dil, G, H, B, rt_row0=nt, depth=depth,
                                     out_split=True).float().cpu()
        rec = got_s[:, :C] + got_s[:, 2 * C:]                     # hi + lo of the split output
        assert torch.equal(got_s[:, :C], got_s[:, C:2 * C])
        assert (rec[:nt] - want_tok).abs().max().item() < 3e-5
        if G:
            real = -(-nt // K)                       # windows that hold at least one token
            err = (got[nt:nt + real] - want[:real, 0]).abs().max().item()
            assert err < 2e-5, ('relay rows', cfg, depth, err)
            assert torch.isfinite(got).all()
        # fp16 (hi, lo) operand layout + fp16-MFMA kernel (v5), where the launch configuration is eligible
        rows_all = nt + W
        if ops.window_attention_f16_ok(rows_all, K, dil, G, H, depth):
            packed = _pack_qkv_f16(torch.cat([qkv_tok, qkv_rt]), H, 0.25 * 1.4426950408889634).to(DEV)
            for tbl in (table.to(DEV), None):
                got5 = ops.window_attention(packed, plan.meta[depth], tbl, nt, W, K, dil, G, H, B, rt_row0=nt,
                                            depth=depth, qkv_f16=True).cpu()
                ref5 = want_tok if tbl is not None else oplan.from_windows(
                    hotformer_ref._sdpa(q, k, v, mask.unsqueeze(1), 0.25).transpose(1, 2).reshape(-1, K + G, C)[:, G:],
                    depth, dil > 1)
                err = (got5[:nt] - ref5).abs().max().item()
                assert err < 3e-5, ('fp16 layout', cfg, depth, G, dil, tbl is None, err)
                if G and tbl is not None:
                    real_w = -(-nt // K)
                    assert (got5[nt:nt + real_w] - want[:real_w, 0]).abs().max().item() < 3e-5
                    assert torch.isfinite(got5).all()
            got5s = ops.window_attention(packed, plan.meta[depth], table.to(DEV), nt, W, K, dil, G, H, B, rt_row0=nt,
                                         depth=depth, qkv_f16=True, out_split=2).float().cpu().view(rows_all, C // 32, 2, 32)
            rec5 = (got5s[:, :, 0] + got5s[:, :, 1]).reshape(rows_all, C)
            assert (rec5[:nt] - want_tok).abs().max().item() < 4e-5
        else:
            assert odepth == 7 and (depth == 5 or G == 0 or K == 64), 'the pyramid depths of the shipped configs must be eligible'
        # no RPE (disable_RPE=True path)
        want0 = hotformer_ref._sdpa(q, k, v, mask.unsqueeze(1), 0.25).transpose(1, 2).reshape(-1, K + G, C)
        for dd in (0, depth):
            got0 = ops.window_attention(torch.cat([qkv_tok, qkv_rt]).to(DEV), plan.meta[depth], None,
                                        nt, W, K, dil, G, H, B, rt_row0=nt, depth=dd).cpu()
            assert (got0[:nt] - oplan.from_windows(want0[:, G:], depth, dil > 1)).abs().max() < 2e-5


@pytest.mark.parametrize('cfg,sizes,depth', [('wild-places', [4096, 30, 2500, 4096], 7),
                                             ('oxford', [4096, 4096], 9)])
def test_relay_attention_init_and_stats(cfg, sizes, depth):
    clouds = [syn.unit_ball_cloud(700 + i, n) for i, n in enumerate(sizes)]
    params, ref, dev, oplan, plan = _plans(clouds, cfg, depth)
    K, H, C, B = params.patch_size, 16, 256, len(sizes)
    g = torch.Generator().manual_seed(4)
    depths = oplan.pyramid_depths
    # relay-token init (masked window mean) and ADaPE statistics
    for d in depths:
        x = torch.randn(plan.n_tokens[d], C, generator=g)
        xw = oplan.pad(x, d).view(-1, K, C).masked_fill(oplan.rt_init_mask[d].unsqueeze(-1), float('nan'))
        want = torch.nanmean(xw, 1)
        got = ops.relay_token_init(x.to(DEV), plan.meta[d], plan.n_windows[d], K).cpu()
        assert torch.allclose(got, want, atol=1e-6, rtol=1e-5)
        if params.ADaPE_mode is not None:
            assert torch.allclose(plan.window_stats[d].cpu(), oplan.window_stats[d], atol=2e-6, rtol=1e-5)
    # RTSA attention core: oracle on padded (B,R,R) masks vs ragged kernel
    rts = {d: torch.randn(plan.n_windows[d], 3 * C, generator=g) for d in depths}
    counts = [oplan.batch_num_windows[d].tolist() for d in depths]
    split = [rts[d].split(counts[j]) for j, d in enumerate(depths)]
    x = hotformer_ref._pad_rows([torch.cat([split[j][b] for j in range(3)]) for b in range(B)])
    q, k, v = x.reshape(B, -1, 3, H, 16).permute(2, 0, 3, 1, 4)
    want = hotformer_ref._sdpa(q, k, v, oplan.rt_attn_mask.unsqueeze(1), 0.25).transpose(1, 2).reshape(B, -1, C)
    got = ops.relay_attention(torch.cat([rts[d] for d in depths]).to(DEV), plan.seq_rows, plan.seq_off,
                              B, H, plan.max_seq_len).cpu()
    lay = plan.layout
    for b in range(B):
        rows = lay['seq_rows'][lay['seq_off'][b]:lay['seq_off'][b + 1]]
        # position of each listed row inside the padded per-cloud sequence of the reference
        pos, p = [], 0
        for j, d in enumerate(depths):
            n = counts[j][b]
            keep = n - (lay['n_pad_windows'][d] if b == B - 1 else 0)
            pos.extend(range(p, p + keep))
            p += n
        assert len(pos) == len(rows)
        err = (got[rows] - want[b, pos]).abs().max().item()
        assert err < 2e-5, (cfg, b, err)
    listed = np.zeros(got.shape[0], bool)
    listed[lay['seq_rows']] = True
    assert torch.all(got[~torch.from_numpy(listed)] == 0)
    # the same attention on the fp16 (hi, lo) operand rows, writing the proj GEMM's split2 operand (hfl_relay_attention_f16_fwd:
    # the inference path's relay-token block): equal to the fp32-operand kernel to the operand's 22 bits; rows of no sequence
    # -- exactly `orphan_rows` -- come out zero
    assert np.array_equal(np.sort(lay['orphan_rows']), np.flatnonzero(~listed))
    x_all = torch.cat([rts[d] for d in depths]).to(DEV)
    packed = _pack_qkv_f16(x_all, H, 0.25 * 1.4426950408889634)
    o2 = ops.relay_attention_f16(packed, plan.seq_rows, plan.seq_off, B, H, plan.max_seq_len, plan.orphan_rows)
    v = o2.float().view(o2.shape[0], C // 32, 2, 32)
    val = (v[:, :, 0] + v[:, :, 1]).reshape(o2.shape[0], C).cpu()
    assert (val - got).abs().max().item() < 3e-5 * max(got.abs().max().item(), 1.0)
    assert torch.all(val[~torch.from_numpy(listed)] == 0)
    # sequences of <= 64 relay tokens take the one-round-trip path of the kernel; the general loop (longer sequences) must agree
    lib = _native.load()
    try:
        assert lib.hfl_set_variant(b'relay_fast', 0) == 0
        o3 = ops.relay_attention_f16(packed, plan.seq_rows, plan.seq_off, B, H, plan.max_seq_len, plan.orphan_rows)
    finally:
        lib.hfl_set_variant(b'relay_fast', 1)
    v3 = o3.float().view(o3.shape[0], C // 32, 2, 32)
    val3 = (v3[:, :, 0] + v3[:, :, 1]).reshape(o3.shape[0], C).cpu()
    assert (val3 - val).abs().max().item() < 2e-5 * max(got.abs().max().item(), 1.0)       # (the output's (hi, lo) pair: 16 bits)


def test_segment_softmax():
    g = torch.Generator().manual_seed(5)
    counts = [700, 1, 2075, 33]
    off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int64)
    for nq in (148, 36, 7):
        s = torch.randn(sum(counts), nq, generator=g) * 3
        want = torch.cat([torch.softmax(s[off[b]:off[b + 1]] * 0.0625, dim=0) for b in range(4)])
        got = ops.segment_softmax_(s.clone().to(DEV), off.to(DEV), 4, 0.0625).cpu()
        assert torch.allclose(got, want, atol=1e-6, rtol=1e-5)


def test_layer_norm_and_fused_add():
    g = torch.Generator().manual_seed(6)
    for n, C in ((1000, 128), (777, 256), (50, 32), (300, 64), (9, 1024), (33, 512), (5, 16)):
        x = torch.randn(n, C, generator=g) * 3 + 0.5
        y = torch.randn(n, C, generator=g)
        w = 1 + 0.1 * torch.randn(C, generator=g)
        b = 0.1 * torch.randn(C, generator=g)
        ab = 0.2 * torch.randn(C, generator=g)
        want = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5)
        got = ops.layer_norm(x.to(DEV), w.to(DEV), b.to(DEV)).cpu()
        assert torch.allclose(got, want, atol=2e-6, rtol=1e-5), (n, C)
        xs = x + y + ab
        xo, h = ops.add_layer_norm(x.to(DEV), y.to(DEV), w.to(DEV), b.to(DEV), add_bias=ab.to(DEV))
        assert torch.allclose(xo.cpu(), xs, atol=1e-6)
        assert torch.allclose(h.cpu(), torch.nn.functional.layer_norm(xs, (C,), w, b, 1e-5), atol=3e-6, rtol=1e-5)
        xo2, h2 = ops.add_layer_norm(x.to(DEV), y.to(DEV), w.to(DEV), b.to(DEV))
        assert torch.allclose(xo2.cpu(), x + y, atol=1e-6)
    x3 = torch.randn(4, 37, 256, generator=g)
    got = ops.layer_norm(x3.to(DEV), torch.ones(256, device=DEV), torch.zeros(256, device=DEV)).cpu()
    assert torch.allclose(got, torch.nn.functional.layer_norm(x3, (256,)), atol=2e-6, rtol=1e-5)


def test_attn_pool_matches_fp64_pooling():
    """hfl_attn_pool (scores -> softmax over the cloud's rows -> weighted sum in one launch) against the pooling of
    models/layers/salsa.py:25-55 in fp64 on ragged clouds: long and short clouds, a cloud of one row, query counts that are
    not multiples of 16 / 64, both channel widths; tolerance of the three-term bf16 split (scores to ~1e-4 absolute).  Also
    against the launches it replaces, and a slice of a wider token matrix as destination."""
    g = torch.Generator().manual_seed(11)
    for C, k, sizes in ((256, 148, [2130, 1977, 1, 33, 2500, 640, 31, 2048]), (256, 36, [65, 70, 3, 64]),
                        (256, 72, [446, 500, 17]), (128, 20, [900, 5, 300])):
        n = sum(sizes)
        x = torch.randn(n, C, generator=g) * 1.3 + 0.2
        q = torch.randn(k, C, generator=g)
        off = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int64)
        scale = C ** -0.5
        want = torch.empty(len(sizes), k, C, dtype=torch.float64)
        for b in range(len(sizes)):
            xb = x[off[b]:off[b + 1]].double()
            want[b] = torch.softmax(q.double() @ xb.t() * scale, dim=-1) @ xb
        xd, qd, od = x.to(DEV), q.to(DEV), off.to(DEV)
        got = ops.attn_pool(xd, od, qd, len(sizes), scale)
        assert got.shape == (len(sizes), k, C)
        err = (got.cpu().double() - want).abs().max().item()
        assert err < 6e-4, (C, k, err)
        # the launches it replaces (fp32 GEMM, segment softmax, padded batched GEMM)
        sc = torch.mm(xd, qd.t())
        ops.segment_softmax_(sc, od, len(sizes), scale)
        nmax = max(sizes)
        old = torch.bmm(ops.pad_rows(sc, od, len(sizes), nmax).transpose(1, 2), ops.pad_rows(xd, od, len(sizes), nmax)) \
            if k % 4 == 0 else None
        if old is not None:
            assert (got - old).abs().max().item() < 6e-4
        # into a slice of a wider (B, K, C) matrix
        wide = torch.full((len(sizes), k + 24, C), 7.0, device=DEV)
        ops.attn_pool(xd, od, qd, len(sizes), scale, out=wide[:, 8:8 + k])
        assert torch.equal(wide[:, 8:8 + k], got) and bool((wide[:, :8] == 7.0).all()) and bool((wide[:, 8 + k:] == 7.0).all())
        # one workgroup per (cloud, query group) -- no workspace -- gives the same result up to the order of the sums
        lib = _native.load()
        one = torch.empty_like(got)
        assert lib.hfl_attn_pool(one.data_ptr(), one.stride(0), xd.data_ptr(), od.data_ptr(), qd.data_ptr(), len(sizes), k, C, n,
                                 float(scale), None, 0, None) == 0
        torch.cuda.synchronize()
        assert (one - got).abs().max().item() < 2e-5


def test_mixer_tail_matches_the_two_linears():
    """hfl_mixer_tail (channel_proj over tokens, row_proj over channels, flatten: models/layers/salsa.py:104-111, with the two
    maps exchanged) against the reference order in fp64."""
    g = torch.Generator().manual_seed(3)
    for b, k, c, ko, d in ((5, 256, 256, 64, 4), (3, 128, 128, 32, 8), (1, 77, 64, 19, 1)):
        x = torch.randn(b, k, c, generator=g)
        wc, bc = torch.randn(ko, k, generator=g) * 0.1, torch.randn(ko, generator=g)
        wr, br = torch.randn(d, c, generator=g) * 0.1, torch.randn(d, generator=g)
        y = torch.nn.functional.linear(x.double().permute(0, 2, 1), wc.double(), bc.double()).permute(0, 2, 1)
        want = torch.nn.functional.linear(y, wr.double(), br.double()).flatten(1)
        got = ops.mixer_tail(x.to(DEV), wc.to(DEV), bc.to(DEV), wr.to(DEV), br.to(DEV)).cpu().double()
        assert got.shape == want.shape
        assert (got - want).abs().max().item() < 2e-5 * want.abs().max().item(), (b, k, c)


def test_layer_norm_relu_f32_and_split2():
    """hfl_layer_norm_relu (norm -> ReLU behind every stem convolution, octformer_layers.py:80-98, in one pass): the fp32 form
    equals relu(LayerNorm) of the two-launch form bit for bit, the split2 form equals split2 of it bit for bit."""
    g = torch.Generator().manual_seed(16)
    for n, C in ((1000, 32), (4099, 64), (777, 128), (9, 256)):
        x = (torch.randn(n, C, generator=g) * 3 + 0.5).to(DEV)
        w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
        b = (0.1 * torch.randn(C, generator=g)).to(DEV)
        two = torch.relu(ops.layer_norm(x, w, b))
        one = ops.layer_norm_relu(x, w, b)
        assert torch.equal(one, two), (n, C)
        assert (one >= 0).all() and (one == 0).any()
        s2 = ops.layer_norm_relu(x, w, b, split2=True)
        assert s2.dtype == torch.bfloat16 and tuple(s2.shape) == (n, 2 * C)
        assert torch.equal(s2.view(torch.int16), ops.split2(two).view(torch.int16)), (n, C)


def test_split_precision_linear_path():
    """bf16 [hi|hi|lo] x [hi|lo|hi] GEMM == fp32 Linear to ~4e-6; producers agree with their fp32 ops."""
    g = torch.Generator().manual_seed(7)

    def unsplit(a3, c):
        a3 = a3.float().cpu()
        assert torch.equal(a3[:, :c], a3[:, c:2 * c])
        return a3[:, :c] + a3[:, 2 * c:]

    for n, cin, cout in ((3000, 256, 768), (1234, 128, 512), (500, 1024, 256)):
        x = torch.randn(n, cin, generator=g)
        w = torch.randn(cout, cin, generator=g) * 0.05
        a3 = ops.split3(x.to(DEV))
        assert (unsplit(a3, cin) - x).abs().max() <= x.abs().max() * 2 ** -15
        y = ops.split_mm(a3, ops.split_weight(w.to(DEV))).cpu().double()
        ref = x.double() @ w.double().t()
        assert ((y - ref).norm() / ref.norm()).item() < 2e-5
    for n, C in ((700, 256), (900, 128)):
        x = torch.randn(n, C, generator=g) * 2
        y = torch.randn(n, C, generator=g)
        w = 1 + 0.1 * torch.randn(C, generator=g)
        b = 0.1 * torch.randn(C, generator=g)
        ab = 0.3 * torch.randn(C, generator=g)
        ln = torch.nn.functional.layer_norm(x, (C,), w, b)
        got = unsplit(ops.layer_norm_split3(x.to(DEV), w.to(DEV), b.to(DEV)), C)
        assert (got - ln).abs().max() < 2e-4 * ln.abs().max()
        xo, h3 = ops.add_layer_norm_split3(x.to(DEV), y.to(DEV), w.to(DEV), b.to(DEV), add_bias=ab.to(DEV))
        xs = x + y + ab
        assert torch.allclose(xo.cpu(), xs, atol=1e-6)
        ln2 = torch.nn.functional.layer_norm(xs, (C,), w, b)
        assert (unsplit(h3, C) - ln2).abs().max() < 2e-4 * ln2.abs().max()
        gl = torch.nn.functional.gelu(x + ab)
        assert (unsplit(ops.bias_gelu_split3(x.to(DEV), ab.to(DEV)), C) - gl).abs().max() < 2e-4 * gl.abs().max()
        assert torch.allclose(ops.add_bias(x.to(DEV), y.to(DEV), ab.to(DEV)).cpu(), xs, atol=1e-6)


def test_window_attention_fused_bias_and_split_output():
    clouds = [syn.unit_ball_cloud(900 + i, n) for i, n in enumerate([3000, 2000])]
    params, ref, dev, oplan, plan = _plans(clouds, 'wild-places', 7)
    K = params.patch_size
    g = torch.Generator().manual_seed(8)
    for depth, H, G in ((5, 8, 0), (4, 16, 1)):
        C = H * 16
        nt, W = plan.n_tokens[depth], plan.n_windows[depth]
        rows = nt + (W if G else 0)
        qkv = torch.randn(rows, 3 * C, generator=g).to(DEV)
        bias = torch.randn(3 * C, generator=g).to(DEV)
        table = (torch.randn(3 * (2 * int(0.8 * K) + 1), H, generator=g) * 0.3).to(DEV)
        want = ops.window_attention(qkv + bias, plan.meta[depth], table, nt, W, K, 1, G, H, 2, rt_row0=nt,
                                    depth=depth)
        got3 = ops.window_attention(qkv, plan.meta[depth], table, nt, W, K, 1, G, H, 2, rt_row0=nt,
                                    depth=depth, qkv_bias=bias, out_split=True).float()
        assert torch.equal(got3[:, :C], got3[:, C:2 * C])
        got = got3[:, :C] + got3[:, 2 * C:]
        assert (got - want).abs().max() < 1e-4 * want.abs().max() + 1e-6
        # split2 rows (operand of hfl_linear_x3): the same hi / lo values, laid out per 32-channel block
        got2 = ops.window_attention(qkv, plan.meta[depth], table, nt, W, K, 1, G, H, 2, rt_row0=nt,
                                    depth=depth, qkv_bias=bias, out_split=2).float().view(rows, C // 32, 2, 32)
        assert torch.equal(got2[:, :, 0].reshape(rows, C), got3[:, :C])
        assert torch.equal(got2[:, :, 1].reshape(rows, C), got3[:, 2 * C:])


def _unsplit2(a2, c):
    a2 = a2.float().cpu().view(a2.shape[0], c // 32, 2, 32)
    return (a2[:, :, 0] + a2[:, :, 1]).reshape(-1, c)


def test_linear_x3_exact_on_small_integers():
    """Layout check of the hand-written split GEMM with exactly representable data (asymmetric operands, M tail,
    every epilogue): integer-valued inputs make x W^T exact in fp32, so any mis-placed fragment shows up as a wrong
    integer, not as rounding noise."""
    g = torch.Generator().manual_seed(21)
    for n, cin, cout in ((300, 64, 128), (129, 32, 256), (1, 96, 128), (515, 256, 384)):
        x = torch.randint(-8, 9, (n, cin), generator=g).float()
        w = torch.randint(-8, 9, (cout, cin), generator=g).float()
        b = torch.randint(-20, 21, (cout,), generator=g).float()
        r = torch.randint(-50, 51, (n, cout), generator=g).float()
        x2, w2 = ops.split2(x.to(DEV)), ops.split2_weight(w.to(DEV))
        assert torch.equal(_unsplit2(x2, cin), x)
        ref = x @ w.t()
        assert torch.equal(ops.linear_x3(x2, w2).cpu(), ref)
        assert torch.equal(ops.linear_x3(x2, w2, bias=b.to(DEV)).cpu(), ref + b)
        assert torch.equal(ops.linear_x3(x2, w2, bias=b.to(DEV), residual=r.to(DEV)).cpu(), ref + b + r)
        buf = r.to(DEV).clone()                                     # residual aliasing the output
        ops.linear_x3(x2, w2, bias=b.to(DEV), residual=buf, out=buf)
        assert torch.equal(buf.cpu(), ref + b + r)
        gl = torch.nn.functional.gelu((ref + b).double()).float()
        got = _unsplit2(ops.linear_x3(x2, w2, bias=b.to(DEV), gelu_split_out=True), cout)
        assert (got - gl).abs().max() <= 2e-5 * gl.abs().max() + 1e-6, (n, cin, cout)


def test_linear_x3_matches_fp64_linear():
    """fp32-equivalent accuracy on real-valued data (three-term bf16 split, fp32 accumulation): <= 1e-5 relative L2
    against fp64, GELU epilogue included; LayerNorm producer of the split2 operand."""
    g = torch.Generator().manual_seed(22)
    for n, cin, cout in ((3000, 256, 768), (1234, 128, 512), (500, 1024, 256), (2049, 256, 1024)):
        x = torch.randn(n, cin, generator=g)
        w = torch.randn(cout, cin, generator=g) * 0.05
        b = torch.randn(cout, generator=g) * 0.1
        ref = x.double() @ w.double().t() + b.double()
        x2, w2 = ops.split2(x.to(DEV)), ops.split2_weight(w.to(DEV))
        assert (_unsplit2(x2, cin) - x).abs().max() <= x.abs().max() * 2 ** -15
        y = ops.linear_x3(x2, w2, bias=b.to(DEV)).cpu().double()
        assert ((y - ref).norm() / ref.norm()).item() < 1e-5
        gl = torch.nn.functional.gelu(ref)
        got = _unsplit2(ops.linear_x3(x2, w2, bias=b.to(DEV), gelu_split_out=True), cout).double()
        assert ((got - gl).norm() / gl.norm()).item() < 1e-5
        # the split2 output itself is a (hi, lo) bf16 pair: representable to 2^-16 of the value
        assert (got - gl).abs().max().item() < 2 ** -15 * max(gl.abs().max().item(), 1.0) + 2e-6
    for n, C in ((700, 256), (900, 128), (33, 1024)):
        x = torch.randn(n, C, generator=g) * 2
        w = 1 + 0.1 * torch.randn(C, generator=g)
        b = 0.1 * torch.randn(C, generator=g)
        ln = torch.nn.functional.layer_norm(x, (C,), w, b)
        got = _unsplit2(ops.layer_norm_split2(x.to(DEV), w.to(DEV), b.to(DEV)), C)
        assert (got - ln).abs().max() < 2e-4 * ln.abs().max()
        # the same (hi, lo) pairs as the K-concatenated producer
        a3 = ops.layer_norm_split3(x.to(DEV), w.to(DEV), b.to(DEV)).float().cpu()
        assert torch.equal(got, a3[:, :C] + a3[:, 2 * C:])


def test_linear_x6_planes_are_exact_and_layout():
    """hfl_linear_x6_pack: the three bf16 planes of a weight sum to the fp32 value EXACTLY (h = RNE(v), m = RNE(v - h),
    l = RNE(v - h - m); 3 x 8 significand bits cover fp32's 24), laid out (3, N, K); and the product of small integers is exact
    with every epilogue (bias, residual in place, row scale) at every row-tile height incl. ragged tails."""
    g = torch.Generator().manual_seed(61)
    w = torch.randn(256, 192, generator=g) * torch.logspace(-6, 3, 192)          # nine decades of magnitude
    w3 = ops.x6_pack(w.to(DEV)).float().cpu()
    assert tuple(w3.shape) == (3, 256, 192)
    w96 = ops.x6_pack(w[:, :96].contiguous().to(DEV)).float().cpu()            # K = 96: padded to 128 with zeros
    assert tuple(w96.shape) == (3, 256, 128) and torch.equal(w96[:, :, :96], w3[:, :, :96]) and (w96[:, :, 96:] == 0).all()
    assert torch.equal(w3[0].double() + w3[1].double() + w3[2].double(), w.double())
    assert (w3[1].abs() <= w3[0].abs() * 2.0 ** -8 + 1e-45).all() and (w3[2].abs() <= w3[0].abs() * 2.0 ** -16 + 1e-45).all()
    lib = ops._native.load()
    lib.hfl_internal_set_x6_mt.argtypes = [ctypes.c_int]
    try:
        for mt in (0, 2, 4, 12, 14):      # tile shapes: chosen per launch, 128 / 256 rows x 128, 64 / 128 rows x 256
            lib.hfl_internal_set_x6_mt(mt)
            for n, cin, cout in ((1, 64, 128), (63, 32, 128), (257, 96, 384), (700, 256, 256), (2100, 128, 512)):
                x = torch.randint(-8, 9, (n, cin), generator=g).float()
                wi = torch.randint(-8, 9, (cout, cin), generator=g).float()
                b = torch.randint(-8, 9, (cout,), generator=g).float()
                r = torch.randint(-64, 65, (n, cout), generator=g).float()
                sc = torch.randint(0, 3, (n,), generator=g).float()
                ref = x @ wi.t()
                w6 = ops.x6_pack(wi.to(DEV))
                assert torch.equal(ops.linear_x6(x.to(DEV), w6).cpu(), ref), (mt, n, cin, cout)
                assert torch.equal(ops.linear_x6(x.to(DEV), w6, bias=b.to(DEV), residual=r.to(DEV)).cpu(), ref + b + r)
                assert torch.equal(ops.linear_x6(x.to(DEV), w6, bias=b.to(DEV), residual=r.to(DEV),
                                                 row_scale=sc.to(DEV)).cpu(), (ref + b) * sc[:, None] + r)
                buf = r.to(DEV).clone()                                   # x += Linear(.): residual aliases out
                ops.linear_x6(x.to(DEV), w6, bias=b.to(DEV), residual=buf, out=buf)
                assert torch.equal(buf.cpu(), ref + b + r)
    finally:
        lib.hfl_internal_set_x6_mt(0)


def test_linear_x6_is_as_accurate_as_the_fp32_library_gemm():
    """The matched-precision Linear (three bf16 planes per operand, six plane products, fp32 accumulation) against fp64 on
    real-valued data, next to torch's fp32 GEMM (hipBLASLt) on the same inputs: relative L2 error not above the library's
    (+ 10 % slack for the accumulation order) on the row counts of the model's step, and <= 1e-6 absolutely everywhere; GELU
    epilogue <= 1e-6 of fp64's exact-erf GELU.  (A short, deep product -- 500 x 1024 -- is the one shape class where the
    library is closer to fp64 than a sequential fp32 dot product, 2.9e-7 against 4.9e-7: it splits K over workgroups there;
    at 70 000 rows it accumulates like this kernel and reads 5.7e-7.)
    Reference: torch.nn.Linear in fp32 (models/octformer_backbone.py:70,91; models/layers/octformer_layers.py:53-59)."""
    g = torch.Generator().manual_seed(62)
    for n, cin, cout in ((30000, 256, 768), (30000, 128, 512), (30000, 1024, 256), (2049, 256, 1024), (70000, 256, 256),
                         (500, 1024, 256), (1234, 96, 128)):
        x = torch.randn(n, cin, generator=g) * 1.3
        w = torch.randn(cout, cin, generator=g) * 0.05
        b = torch.randn(cout, generator=g) * 0.1
        ref = x.double() @ w.double().t() + b.double()
        xd, w6 = x.to(DEV), ops.x6_pack(w.to(DEV))
        e6 = ((ops.linear_x6(xd, w6, bias=b.to(DEV)).cpu().double() - ref).norm() / ref.norm()).item()
        e32 = ((torch.nn.functional.linear(xd, w.to(DEV), b.to(DEV)).cpu().double() - ref).norm() / ref.norm()).item()
        print('x6 %.2e  fp32 library %.2e  (M %d K %d N %d)' % (e6, e32, n, cin, cout))
        assert e6 < 1e-6, (e6, e32, n, cin, cout)
        if n >= 2000:
            assert e6 <= 1.1 * e32, (e6, e32, n, cin, cout)
        gl = torch.nn.functional.gelu(ref)
        eg = ((ops.linear_x6(xd, w6, bias=b.to(DEV), gelu=True).cpu().double() - gl).norm() / gl.norm()).item()
        assert eg < 1e-6, eg


def test_linear_x6_grouped_gather_matches_per_block_products():
    """hfl_linear_x6_grouped_gather (the per-tap products of a live-tap octree convolution at matched precision): one launch
    over row tiles with different weight blocks and a row index per operand row; exact on small integers, <= 3e-7 relative L2
    of fp64 on real data; ragged tile heights, empty blocks, out_features = 64 on 128-row zero-padded blocks, Cin = 32."""
    g = torch.Generator().manual_seed(63)
    for cin, cout in ((64, 64), (128, 128), (32, 64), (64, 128), (128, 256)):
        npad = max(cout, 128)
        sizes = [0, 1, 127, 128, 129, 700, 0, 33, 2049]                    # rows per block ("tap")
        nb = len(sizes)
        edges = [0]
        for n in sizes:
            edges.append(edges[-1] + n)
        tiles = [(a, min(128, edges[k + 1] - a), k * npad) for k in range(nb) for a in range(edges[k], edges[k + 1], 128)]
        tiles_t = torch.tensor(tiles, dtype=torch.int32, device=DEV)
        n_src = 1500
        src = torch.randint(0, n_src, (edges[-1],), generator=g, dtype=torch.int32)
        for integer in (True, False):
            if integer:
                x = torch.randint(-8, 9, (n_src, cin), generator=g).float()
                w = torch.randint(-8, 9, (nb, cout, cin), generator=g).float()
            else:
                x = torch.randn(n_src, cin, generator=g)
                w = torch.randn(nb, cout, cin, generator=g) * 0.1
            wp = torch.cat([w, w.new_zeros(nb, npad - cout, cin)], 1) if npad > cout else w
            w3 = ops.x6_pack(wp.reshape(nb * npad, cin).contiguous().to(DEV))
            got = ops.linear_x6_grouped_gather(x.to(DEV), src.to(DEV), w3, tiles_t, cout).cpu()
            xs = x[src.long()]
            ref = torch.cat([xs[edges[k]:edges[k + 1]].double() @ w[k].double().t() for k in range(nb)], 0)
            if integer:
                assert torch.equal(got.double(), ref), (cin, cout)
            else:
                assert ((got.double() - ref).norm() / ref.norm()).item() < 3e-7, (cin, cout)


def test_linear_x3_grouped_matches_per_block_products():
    """One launch over row tiles with different weight blocks (hfl_linear_x3_grouped, the per-tap products of a live-tap
    octree convolution): exact on small integers, <= 1e-5 relative L2 on real data; ragged tile heights (1..128 rows), empty
    blocks, out_features = 64 on 128-row zero-padded blocks."""
    g = torch.Generator().manual_seed(51)
    for cin, cout in ((64, 64), (128, 128), (32, 64), (64, 128), (128, 256)):
        npad = max(cout, 128)
        sizes = [0, 1, 127, 128, 129, 700, 0, 33, 2049]                    # rows per block ("tap")
        nb = len(sizes)
        edges = [0]
        for n in sizes:
            edges.append(edges[-1] + n)
        tiles = [(a, min(128, edges[k + 1] - a), k * npad) for k in range(nb) for a in range(edges[k], edges[k + 1], 128)]
        tiles_t = torch.tensor(tiles, dtype=torch.int32, device=DEV)
        for integer in (True, False):
            if integer:
                x = torch.randint(-3, 4, (edges[-1], cin), generator=g).float()
                w = torch.randint(-3, 4, (nb, cout, cin), generator=g).float()
            else:
                x = torch.randn(edges[-1], cin, generator=g)
                w = torch.randn(nb, cout, cin, generator=g) * 0.1
            wp = torch.cat([w, torch.zeros(nb, npad - cout, cin)], 1) if npad > cout else w
            got = ops.linear_x3_grouped(ops.split2(x.to(DEV)), ops.split2(wp.reshape(nb * npad, cin).to(DEV)), tiles_t,
                                        cout).cpu()
            ref = torch.cat([x[edges[k]:edges[k + 1]].double() @ w[k].double().t() for k in range(nb)], 0)
            if integer:
                assert torch.equal(got, ref.float()), (cin, cout)
            else:
                assert ((got.double() - ref).norm() / ref.norm()).item() < 1e-5, (cin, cout)
            # the gather done by the GEMM's tile loader (hfl_linear_x3_grouped_gather): row m = source[src[m]], same bits as
            # the GEMM over the materialised rows
            n_src = 3000
            source = torch.randint(-3, 4, (n_src, cin), generator=g).float() if integer else torch.randn(n_src, cin, generator=g)
            src = torch.randint(0, n_src, (edges[-1],), generator=g, dtype=torch.int32)
            w2 = ops.split2(wp.reshape(nb * npad, cin).to(DEV))
            mat = ops.linear_x3_grouped(ops.split2(source[src.long()].to(DEV)), w2, tiles_t, cout)
            fused = ops.linear_x3_grouped_gather(ops.split2(source.to(DEV)), src.to(DEV), w2, tiles_t, cout)
            assert torch.equal(mat, fused), (cin, cout, integer)


def test_wgrad_x3_matches_fp64_and_is_reproducible():
    """dW = dy^T x and db = column sums of dy from split2 operands (hfl_wgrad_x3): exact on small integers (every product
    and partial sum is representable), <= 1e-5 relative L2 against fp64 on real data, ragged row counts (not a multiple of
    the 32-row step, fewer rows than one step, many slabs), bitwise equal across runs (fixed slab order, no atomics)."""
    g = torch.Generator().manual_seed(31)
    for m, n, k in ((1, 128, 128), (31, 256, 128), (32, 128, 256), (1000, 128, 128), (4099, 256, 1024),
                    (70001, 1024, 256), (20000, 384, 128), (9000, 256, 256)):
        dy = torch.randint(-4, 5, (m, n), generator=g).float()
        x = torch.randint(-4, 5, (m, k), generator=g).float()
        dw, db = ops.wgrad_x3(ops.split2(dy.to(DEV)), ops.split2(x.to(DEV)), with_bias=True)
        assert torch.equal(dw.cpu(), (dy.double().t() @ x.double()).float()), (m, n, k)
        assert torch.equal(db.cpu(), dy.sum(0)), (m, n, k)
        dy = torch.randn(m, n, generator=g) * 0.3
        x = torch.randn(m, k, generator=g) * 2
        dys, xs = ops.split2(dy.to(DEV)), ops.split2(x.to(DEV))
        dw, db = ops.wgrad_x3(dys, xs, with_bias=True)
        ref = dy.double().t() @ x.double()
        assert ((dw.cpu().double() - ref).norm() / ref.norm()).item() < 1e-5, (m, n, k)
        assert (db.cpu().double() - dy.double().sum(0)).abs().max().item() < 1e-5 * (dy.abs().sum(0).max().item() + 1)
        dw2, none = ops.wgrad_x3(dys, xs, with_bias=False)
        assert none is None and torch.equal(dw, dw2)


def test_mlp_x3_matches_fp64_autograd():
    """fc2(gelu(fc1(h))) with GELU and its derivative fused into the GEMM epilogues (autograd.MlpX3Fn): output and all five
    gradients against torch autograd in fp64 (exact erf GELU), <= 2e-5 relative L2; ragged row count."""
    from hotformerloc_amd import autograd as ag
    g = torch.Generator().manual_seed(41)
    for m, c in ((1000, 128), (4133, 256)):
        h = torch.randn(m, c, generator=g)
        w1 = torch.randn(4 * c, c, generator=g) * 0.06
        b1 = torch.randn(4 * c, generator=g) * 0.1
        w2 = torch.randn(c, 4 * c, generator=g) * 0.03
        b2 = torch.randn(c, generator=g) * 0.1
        dy = torch.randn(m, c, generator=g)
        ref_in = [t.double().requires_grad_(True) for t in (h, w1, b1, w2, b2)]
        ref = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(ref_in[0], ref_in[1], ref_in[2])),
                                         ref_in[3], ref_in[4])
        ref.backward(dy.double())
        dev_in = [t.to(DEV).requires_grad_(True) for t in (h, w1, b1, w2, b2)]
        out = ag.mlp_x3(*dev_in)
        out.backward(dy.to(DEV))
        assert ((out.detach().cpu().double() - ref.detach()).norm() / ref.detach().norm()).item() < 2e-5
        for a, b, name in zip(dev_in, ref_in, ('dh', 'dw1', 'db1', 'dw2', 'db2')):
            err = ((a.grad.cpu().double() - b.grad).norm() / b.grad.norm()).item()
            assert err < 2e-5, (m, c, name, err)


def test_ln_mlp_residual_x3_matches_fp64_autograd():
    """x + fc2(gelu(fc1(LN(x)))) as one autograd Function (autograd.LnMlpResidualX3Fn): output and all seven gradients
    against torch autograd in fp64, <= 3e-5 relative L2."""
    from hotformerloc_amd import autograd as ag
    g = torch.Generator().manual_seed(43)
    for m, c in ((777, 128), (3001, 256)):
        x = torch.randn(m, c, generator=g) * 1.5
        gamma = 1 + 0.2 * torch.randn(c, generator=g)
        beta = 0.1 * torch.randn(c, generator=g)
        w1 = torch.randn(4 * c, c, generator=g) * 0.06
        b1 = torch.randn(4 * c, generator=g) * 0.1
        w2 = torch.randn(c, 4 * c, generator=g) * 0.03
        b2 = torch.randn(c, generator=g) * 0.1
        dy = torch.randn(m, c, generator=g)
        F = torch.nn.functional
        r = [t.double().requires_grad_(True) for t in (x, gamma, beta, w1, b1, w2, b2)]
        ref = r[0] + F.linear(F.gelu(F.linear(F.layer_norm(r[0], (c,), r[1], r[2], 1e-5), r[3], r[4])), r[5], r[6])
        ref.backward(dy.double())
        d = [t.to(DEV).requires_grad_(True) for t in (x, gamma, beta, w1, b1, w2, b2)]
        out = ag.ln_mlp_residual_x3(d[0], d[1], d[2], 1e-5, d[3], d[4], d[5], d[6])
        out.backward(dy.to(DEV))
        assert ((out.detach().cpu().double() - ref.detach()).norm() / ref.detach().norm()).item() < 2e-5
        for a, b, name in zip(d, r, ('dx', 'dgamma', 'dbeta', 'dw1', 'db1', 'dw2', 'db2')):
            err = ((a.grad.cpu().double() - b.grad).norm() / b.grad.norm()).item()
            assert err < 3e-5, (m, c, name, err)


def test_window_attention_backward_matches_autograd():
    """dQ, dK, dV and the RPE-table gradient of the HIP backward against torch autograd over the
    oracle's materialised formulation (both window kinds, dilation, relay token)."""
    from hotformerloc_amd import autograd as ag
    clouds = [syn.unit_ball_cloud(1200 + i, n) for i, n in enumerate([3000, 40, 2200])]
    for cfg in ('wild-places', 'cs-wild-places'):
        params, ref, dev, oplan, plan = _plans(clouds, cfg, 7)
        K, D = params.patch_size, params.dilation
        g = torch.Generator().manual_seed(9)
        B = len(clouds)
        for depth, H, G, dil in ((5, 8, 0, 1), (5, 8, 0, D), (4, 16, 1, 1), (2, 16, 1, 1)):
            C = H * 16
            nt, W = plan.n_tokens[depth], plan.n_windows[depth]
            qkv_tok = torch.randn(nt, 3 * C, generator=g, requires_grad=True)
            qkv_rt = torch.randn(W, 3 * C, generator=g, requires_grad=True)
            bnd = int(0.8 * K * dil ** 0.5)
            table = (torch.randn(3 * (2 * bnd + 1), H, generator=g) * 0.5).requires_grad_()
            xw = oplan.to_windows(qkv_tok, depth, dil > 1)
            if G:
                xw = torch.cat([qkv_rt.unsqueeze(1), xw], 1)
            q, k, v = xw.reshape(-1, K + G, 3, H, 16).permute(2, 0, 3, 1, 4)
            if dil > 1:
                mask, pos = oplan.dilate_mask[depth], oplan.dilate_pos[depth]
            else:
                mask, pos = (oplan.hat_mask[depth] if G else oplan.patch_mask[depth]), oplan.rel_pos[depth]
            rpe = hotformer_ref.rpe_bias(table, pos, K, dil)
            if G:
                rpe = torch.nn.functional.pad(rpe, (G, 0, G, 0))
            o = hotformer_ref._sdpa(q, k, v, mask.unsqueeze(1) + rpe, 0.25).transpose(1, 2).reshape(-1, K + G, C)
            o_tok = oplan.from_windows(o[:, G:], depth, dil > 1)
            real = -(-nt // K)
            wt = torch.randn(nt, C, generator=g)
            wr = torch.randn(W, C, generator=g)
            loss = (o_tok * wt).sum()
            if G:
                loss = loss + (o[:real, 0] * wr[:real]).sum()
            loss.backward()
            # HIP path
            qd = torch.cat([qkv_tok.detach(), qkv_rt.detach()]).to(DEV).requires_grad_()
            td = table.detach().to(DEV).requires_grad_()
            od = ag.window_attention(qd, td, plan.meta[depth], n_tokens=nt, n_windows=W, patch_size=K,
                                     dilation=dil, n_relay=G, n_heads=H, batch_size=B, rt_row0=nt, depth=depth)
            wd = torch.cat([wt, wr]).to(DEV)
            if G:
                wd[nt + real:] = 0
                ld = (od * wd).sum()
            else:
                ld = (od[:nt] * wd[:nt]).sum()
            ld.backward()
            gq = qd.grad.cpu()
            scale = qkv_tok.grad.abs().max().item()
            assert (gq[:nt] - qkv_tok.grad).abs().max().item() < 3e-5 * max(scale, 1), (cfg, depth, G, dil)
            if G:
                assert (gq[nt:nt + real] - qkv_rt.grad[:real]).abs().max().item() < 3e-5 * max(scale, 1)
            tscale = table.grad.abs().max().item()
            assert (td.grad.cpu() - table.grad).abs().max().item() < 1e-4 * max(tscale, 1), (cfg, depth, G, dil)
            # the table gradient is reproducible: partial tables per grid column, fixed-order sum (no float atomics)
            qd2 = qd.detach().clone().requires_grad_()
            td2 = table.detach().to(DEV).requires_grad_()
            od2 = ag.window_attention(qd2, td2, plan.meta[depth], n_tokens=nt, n_windows=W, patch_size=K,
                                      dilation=dil, n_relay=G, n_heads=H, batch_size=B, rt_row0=nt, depth=depth)
            ((od2 * wd).sum() if G else (od2[:nt] * wd[:nt]).sum()).backward()
            rows_ok = nt + real if G else nt                     # rows of padding windows are never written
            assert torch.equal(td2.grad, td.grad), (cfg, depth, G, dil)
            assert torch.equal(qd2.grad[:rows_ok], qd.grad[:rows_ok]), (cfg, depth, G, dil)
            # hfl_window_attention_bwd_split2: the gradient written as the split2 GEMM operand = hfl_split2 of the f32 one, bitwise
            dsp = torch.zeros((qd.shape[0], 6 * C), dtype=torch.bfloat16, device=DEV)
            dtab = torch.zeros_like(td)
            desc = ag._desc(nt, W, K, dil, G, H, B, nt, depth)
            ag._window_attention_bwd(dsp, dtab, qd.detach(), wd if G else torch.cat([wd[:nt], torch.zeros_like(wd[nt:])]),
                                     plan.meta[depth], td.detach(), desc, split=True)
            assert torch.equal(dsp[:rows_ok].view(torch.int16), ops.split2(qd.grad[:rows_ok].contiguous()).view(torch.int16))
            assert torch.equal(dtab, td.grad), (cfg, depth, G, dil)
            # the table gradient on the matrix cores (depth <= 5: what ran above) against the LDS scatter-add of deeper levels
            lib = _native.load()
            try:
                assert lib.hfl_set_variant(b'window_bwd_rt', 0) == 0
                dq0 = torch.zeros((qd.shape[0], 3 * C), device=DEV)
                dt0 = torch.zeros_like(td)
                ag._window_attention_bwd(dq0, dt0, qd.detach(), wd if G else torch.cat([wd[:nt], torch.zeros_like(wd[nt:])]),
                                         plan.meta[depth], td.detach(), desc)
            finally:
                lib.hfl_set_variant(b'window_bwd_rt', -1)
            assert (dt0 - td.grad).abs().max().item() < 2e-5 * max(tscale, 1), (cfg, depth, G, dil)
            assert (dq0[:rows_ok] - qd.grad[:rows_ok]).abs().max().item() < 2e-5 * max(scale, 1)


def test_gather_and_relay_init_backward():
    from hotformerloc_amd import autograd as ag
    clouds = [syn.unit_ball_cloud(1300, 2500), syn.forest_cloud(1301, 1800)]
    ref = oracle_octree(clouds, 7)
    dev = build_batch_octree(clouds, 7, 2, DEV)
    g = torch.Generator().manual_seed(10)
    for depth, C, kernel, stride in ((6, 64, '333', 1), (6, 32, '222', 2), (7, 3, '333', 1)):
        n = int(ref.nnum_nempty[depth])
        x = torch.randn(n, C, generator=g, requires_grad=True)
        col = onn.octree_gather(x, ref.get_neigh(depth, kernel, stride, True)).flatten(1)
        wgt = torch.randn(col.shape, generator=g)
        (col * wgt).sum().backward()
        xd = x.detach().to(DEV).requires_grad_()
        cd = ag.octree_gather(xd, dev.get_neigh(depth, kernel, stride, True).contiguous())
        (cd * wgt.to(DEV)).sum().backward()
        assert torch.allclose(xd.grad.cpu(), x.grad, atol=1e-5, rtol=1e-5), (depth, C, kernel)
    params, ref, dev, oplan, plan = _plans(clouds, 'wild-places', 7)
    for d in oplan.pyramid_depths:
        x = torch.randn(plan.n_tokens[d], 256, generator=g, requires_grad=True)
        xw = oplan.pad(x, d).view(-1, 48, 256).masked_fill(oplan.rt_init_mask[d].unsqueeze(-1), float('nan'))
        rt = torch.nanmean(xw, 1)
        wgt = torch.randn(rt.shape, generator=g)
        (rt * wgt).sum().backward()
        xd = x.detach().to(DEV).requires_grad_()
        rd = ag.relay_token_init(xd, plan.meta[d], plan.n_windows[d], 48)
        (rd * wgt.to(DEV)).sum().backward()
        assert torch.allclose(xd.grad.cpu(), x.grad, atol=1e-6, rtol=1e-5)

def test_relay_attention_backward():
    from hotformerloc_amd import autograd as ag
    clouds = [syn.unit_ball_cloud(1500 + i, n) for i, n in enumerate([4096, 50, 3000])]
    params, ref, dev, oplan, plan = _plans(clouds, 'wild-places', 7)
    g = torch.Generator().manual_seed(12)
    rows = sum(plan.n_windows[d] for d in plan.pyramid_depths)
    qkv = torch.randn(rows, 768, generator=g)
    wgt = torch.randn(rows, 256, generator=g)
    a = qkv.clone().to(DEV).requires_grad_()
    (ag.relay_attention_torch(a, plan, 16) * wgt.to(DEV)).sum().backward()
    b = qkv.clone().to(DEV).requires_grad_()
    out = ag.relay_attention(b, plan, 16)
    (out * wgt.to(DEV)).sum().backward()
    assert torch.allclose(out.detach(), ag.relay_attention_torch(a.detach(), plan, 16), atol=2e-5)
    scale = a.grad.abs().max().item()
    assert (a.grad - b.grad).abs().max().item() < 3e-5 * max(scale, 1.0)


def test_layer_norm_backward_matches_torch_autograd():
    """hfl_layer_norm_bwd (dx, dgamma, dbeta; statistics recomputed from x) against torch autograd in float64."""
    from hotformerloc_amd import autograd as ag
    g = torch.Generator().manual_seed(41)
    for n, C in ((1000, 128), (777, 256), (50, 32), (300, 64), (9, 1024), (33, 512), (5, 16), (70001, 256)):
        x = (torch.randn(n, C, generator=g) * 3 + 0.5)
        w = 1 + 0.1 * torch.randn(C, generator=g)
        b = 0.1 * torch.randn(C, generator=g)
        dy = torch.randn(n, C, generator=g)
        xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
        torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5).backward(dy.double())
        xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
        y = ag.layer_norm(xd, wd, bd, 1e-5)
        assert torch.allclose(y.detach().cpu(), torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5), atol=3e-6, rtol=1e-5)
        y.backward(dy.to(DEV))
        assert (xd.grad.cpu().double() - xr.grad).abs().max().item() < 2e-5 * max(xr.grad.abs().max().item(), 1.0), (n, C)
        for got, want in ((wd.grad, wr.grad), (bd.grad, br.grad)):
            err = (got.cpu().double() - want).abs().max().item()
            assert err < 1e-5 * max(want.abs().max().item(), 1.0) * max(1.0, (n / 1000) ** 0.5), (n, C, err)
    # determinism: fixed-order partial sums
    xd = torch.randn(5000, 256, generator=g).to(DEV); dyd = torch.randn(5000, 256, generator=g).to(DEV)
    wd = torch.ones(256, device=DEV)
    a = ops.layer_norm_bwd(dyd, xd, wd)
    b2 = ops.layer_norm_bwd(dyd, xd, wd)
    assert all(torch.equal(u, v) for u, v in zip(a, b2))


def _mlp_ref(x, gamma, beta, eps, w1, b1, w2, b2):
    x = x.double()
    h = torch.nn.functional.layer_norm(x, (x.shape[1],), gamma.double(), beta.double(), eps)
    h = torch.nn.functional.gelu(h @ w1.double().t() + b1.double())
    return x + h @ w2.double().t() + b2.double()


@pytest.mark.parametrize('C', [256, 128])
def test_ln_mlp_fused_matches_fp64_and_the_unfused_launches(C):
    """x + fc2(gelu(fc1(LN(x)))) in one launch (hfl_ln_mlp_fused: hidden activation kept in registers) against fp64 and
    against the three launches it replaces (LayerNorm -> split2, fc1 + GELU, fc2 + residual), over row counts that hit
    every tail: single row, partial 16-row tiles, fewer tiles than workgroups, several passes per workgroup."""
    g = torch.Generator().manual_seed(77 + C)
    w1 = (torch.randn(4 * C, C, generator=g) * 0.05).to(DEV)
    w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).to(DEV)
    b1 = (torch.randn(4 * C, generator=g) * 0.1).to(DEV)
    b2 = (torch.randn(C, generator=g) * 0.1).to(DEV)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    pack = ops.mlp_fused_pack(w1, w2)
    w1s, w2s = ops.split2_weight(w1), ops.split2_weight(w2)
    for n in (1, 15, 16, 17, 129, 2092, 4099, 40000, 70001):
        x = (torch.randn(n, C, generator=g) * 2).to(DEV)
        got = ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2)
        ref = _mlp_ref(x.cpu(), gamma.cpu(), beta.cpu(), 1e-5, w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu())
        err = ((got.cpu().double() - ref).norm() / ref.norm()).item()
        assert err < 1e-5, (C, n, err)
        h2 = ops.layer_norm_split2(x, gamma, beta, 1e-5)
        unf = ops.linear_x3(ops.linear_x3(h2, w1s, bias=b1, gelu_split_out=True), w2s, bias=b2, residual=x)
        assert (got - unf).abs().max().item() <= 2e-5 * ref.abs().max().item(), (C, n)
        assert torch.equal(got, ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2))     # deterministic


@pytest.mark.gpu
def test_ln_mlp_fused_mixer_shape_hidden_equals_channels():
    """The same launch with hidden = C = 256 (hfl_ln_mlp_fused_h: a FeatureMixerLayer of the pooling head,
    models/layers/salsa.py:58-75, mlp_ratio 1) against fp64 and the three launches it replaces; the bench's 8192 token rows
    (hidden split over 4 workgroups per row set), ragged counts, and shapes the entry point must refuse."""
    C = 256
    g = torch.Generator().manual_seed(31)
    w1 = (torch.randn(C, C, generator=g) * 0.06).to(DEV)
    w2 = (torch.randn(C, C, generator=g) * 0.06).to(DEV)
    b1 = (torch.randn(C, generator=g) * 0.1).to(DEV)
    b2 = (torch.randn(C, generator=g) * 0.1).to(DEV)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    assert ops.mlp_fused_shape_ok(256, 256) and ops.mlp_fused_shape_ok(256, 1024) and ops.mlp_fused_shape_ok(128, 512)
    assert not ops.mlp_fused_shape_ok(128, 128) and not ops.mlp_fused_shape_ok(256, 512) and not ops.mlp_fused_shape_ok(64, 256)
    pack = ops.mlp_fused_pack(w1, w2)
    w1s, w2s = ops.split2_weight(w1), ops.split2_weight(w2)
    for n in (1, 17, 300, 8192, 40001):
        x = (torch.randn(n, C, generator=g) * 2).to(DEV)
        got = ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2)
        ref = _mlp_ref(x.cpu(), gamma.cpu(), beta.cpu(), 1e-5, w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu())
        assert ((got.cpu().double() - ref).norm() / ref.norm()).item() < 1e-5, n
        h2 = ops.layer_norm_split2(x, gamma, beta, 1e-5)
        unf = ops.linear_x3(ops.linear_x3(h2, w1s, bias=b1, gelu_split_out=True), w2s, bias=b2, residual=x)
        assert (got - unf).abs().max().item() <= 2e-5 * ref.abs().max().item(), n
        assert torch.equal(got, ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2))


@pytest.mark.gpu
@pytest.mark.parametrize('C,n', [(256, 68167), (256, 33000), (128, 70001)])
def test_ln_mlp_fused_tail_split_equals_whole_passes(C, n):
    """Rows left over after the last whole round of passes (68167 rows at C = 256 are 2.08 rounds of the grid's 32768) are
    computed with the hidden dimension split over several workgroups per row set and summed in a fixed order
    (hfl_ln_mlp_fused_ws): same values as the whole-pass schedule up to the summation order of fc2's partial sums, the rows of
    the whole passes bitwise equal, deterministic, and the workspace query says when the split applies."""
    from hotformerloc_amd import _native
    lib = _native.load()
    g = torch.Generator().manual_seed(C + n)
    w1 = (torch.randn(4 * C, C, generator=g) * 0.05).to(DEV)
    w2 = (torch.randn(C, 4 * C, generator=g) * 0.05).to(DEV)
    b1 = (torch.randn(4 * C, generator=g) * 0.1).to(DEV)
    b2 = (torch.randn(C, generator=g) * 0.1).to(DEV)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    pack = ops.mlp_fused_pack(w1, w2)
    x = (torch.randn(n, C, generator=g) * 2).to(DEV)
    ws = int(lib.hfl_ln_mlp_fused_workspace(n, C))
    assert ws > 0, 'this shape has left-over rows: the split must apply'
    split = ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2)
    assert torch.equal(split, ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2))
    try:
        lib.hfl_set_variant(b'tail_split', 0)
        assert int(lib.hfl_ln_mlp_fused_workspace(n, C)) == 0
        whole = ops.ln_mlp_fused(x, gamma, beta, 1e-5, pack, b1, b2)
    finally:
        lib.hfl_set_variant(b'tail_split', 1)
    ref = _mlp_ref(x.cpu(), gamma.cpu(), beta.cpu(), 1e-5, w1.cpu(), b1.cpu(), w2.cpu(), b2.cpu())
    for got in (split, whole):
        assert ((got.cpu().double() - ref).norm() / ref.norm()).item() < 1e-5
    assert (split - whole).abs().max().item() <= 2e-5 * ref.abs().max().item()
    tail_rows = ws // (C * 4)                              # parts * tail rows; at least one part
    assert 0 < tail_rows


@pytest.mark.gpu
def test_ln_mlp_fused_layout_exact_on_small_integers():
    """Fragment / stage layout check with exactly representable data: an identity-like LayerNorm (constant rows are avoided;
    gamma = 1, beta = 0 on rows whose statistics are exact is not available, so the LayerNorm is bypassed by feeding rows with
    mean 0 and unit variance built from +-1 entries) and integer weights make every product exact."""
    C = 256
    g = torch.Generator().manual_seed(5)
    # rows of +-1 with equal counts: mean 0, variance 1 exactly -> LN(x) = x / sqrt(1 + eps) with eps = 0
    x = torch.ones(300, C)
    x[:, ::2] = -1
    perm = torch.stack([torch.randperm(C, generator=g) for _ in range(300)])
    x = torch.gather(x, 1, perm)
    w1 = torch.randint(-2, 3, (4 * C, C), generator=g).float()
    w2 = torch.randint(-2, 3, (C, 4 * C), generator=g).float()
    b1 = torch.randint(-3, 4, (4 * C,), generator=g).float()
    b2 = torch.randint(-3, 4, (C,), generator=g).float()
    pack = ops.mlp_fused_pack(w1.to(DEV), w2.to(DEV))
    got = ops.ln_mlp_fused(x.to(DEV), torch.ones(C, device=DEV), torch.zeros(C, device=DEV), 0.0, pack, b1.to(DEV),
                           b2.to(DEV)).cpu()
    ref = _mlp_ref(x, torch.ones(C), torch.zeros(C), 0.0, w1, b1, w2, b2)
    # GELU of integers is not an integer: compare to fp64 at the (hi, lo) bf16 resolution of the hidden activation
    assert (got.double() - ref).abs().max().item() < 3e-5 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize('rows,c', [(1, 128), (17, 128), (5000, 128), (70000, 128), (33, 256), (4099, 256), (40000, 256)])
def test_ln_qkv_fused_matches_fp64_and_the_two_launch_form(rows, c):
    """hfl_ln_qkv_fused (LayerNorm -> qkv projection -> fp16 (hi, lo) attention operand rows in one launch) against fp64
    LayerNorm + Linear, and against hfl_layer_norm_split2 + hfl_linear_x3_qkv, whose arithmetic it shares (same split
    operands, same fp16 split of the result): ragged row counts, several workgroup passes, both channel widths."""
    g = torch.Generator(device='cuda').manual_seed(rows + c)
    x = torch.randn(rows, c, device='cuda', generator=g) * 1.7 + 0.4
    gamma = torch.rand(c, device='cuda', generator=g) + 0.5
    beta = torch.randn(c, device='cuda', generator=g) * 0.1
    w = torch.randn(3 * c, c, device='cuda', generator=g) * 0.06
    b = torch.randn(3 * c, device='cuda', generator=g) * 0.1
    qs = 16 ** -0.5 * 1.4426950408889634

    def decode(buf):           # per head 64 B = [16 x hi | 16 x lo] fp16
        h = buf.view(torch.float16).reshape(rows, 3, c // 16, 2, 16).float()
        return (h[:, :, :, 0] + h[:, :, :, 1]).reshape(rows, 3 * c)

    fused = decode(ops.ln_qkv_fused(x, gamma, beta, 1e-5, ops.qkv_fused_pack(w), b, qs))
    two = decode(ops.linear_x3_qkv(ops.layer_norm_split2(x, gamma, beta, 1e-5), ops.split2_weight(w), b, qs))
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x.double(), (c,), gamma.double(), beta.double(), 1e-5),
                                     w.double(), b.double())
    ref[:, :c] *= qs
    assert torch.isfinite(fused).all()
    assert ((fused.double() - ref).norm() / ref.norm()).item() < 1e-5
    assert (fused - two).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.gpu
def test_window_attention_multi_launch_equals_single_launches():
    """hfl_window_attention_fwd_multi: the three pyramid levels of a Wild-Places batch (same K, relay token, heads; different
    depths, row counts and expanded tables) in ONE launch against one launch each -- the same windows through the same kernel,
    bitwise equal; a problem with another shape in the list falls back to single launches and is equal as well."""
    sizes = [4096, 30, 2500, 4096]
    clouds = [syn.unit_ball_cloud(700 + i, n) for i, n in enumerate(sizes)]
    params, ref, dev, oplan, plan = _plans(clouds, 'wild-places', 7)
    K = params.patch_size
    g = torch.Generator(device='cuda').manual_seed(9)
    probs = []
    for depth in (4, 3, 2):
        H, C = 16, 256
        nt, W = plan.n_tokens[depth], plan.n_windows[depth]
        x = torch.randn(nt + W, C, device='cuda', generator=g)
        w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.06
        b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
        qkv = ops.linear_x3_qkv(ops.split2(x), ops.split2_weight(w), b, 0.25 * 1.4426950408889634)
        table = torch.randn(3 * (2 * int(0.8 * K) + 1), H, device='cuda', generator=g) * 0.3
        probs.append(dict(qkv=qkv, tok_meta=plan.meta[depth], rpe_table=table, n_tokens=nt, n_windows=W, patch_size=K,
                          dilation=1, n_relay=1, n_heads=H, batch_size=len(sizes), rt_row0=nt, depth=depth))
    single = [ops.window_attention(p['qkv'], p['tok_meta'], p['rpe_table'], p['n_tokens'], p['n_windows'], K, 1, 1, 16,
                                   len(sizes), rt_row0=p['n_tokens'], depth=p['depth'], out_split=2, qkv_f16=True)
              for p in probs]
    multi = ops.window_attention_multi([dict(p) for p in probs])
    for p, a, m in zip(probs, single, multi):
        used = p['n_tokens'] + (-(-p['n_tokens'] // K))          # token rows + relay rows of windows that hold tokens
        assert torch.equal(a[:used].view(torch.int16), m[:used].view(torch.int16)), p['depth']
    # depth 5 without relay token (another shape): the call must still give every problem its own result
    d5 = 5
    nt5, W5 = plan.n_tokens[d5], plan.n_windows[d5]
    x5 = torch.randn(nt5, 128, device='cuda', generator=g)
    qkv5 = ops.linear_x3_qkv(ops.split2(x5), ops.split2_weight(torch.randn(384, 128, device='cuda', generator=g) * 0.08),
                             torch.zeros(384, device='cuda'), 0.25 * 1.4426950408889634)
    t5 = torch.randn(3 * (2 * int(0.8 * K) + 1), 8, device='cuda', generator=g) * 0.3
    p5 = dict(qkv=qkv5, tok_meta=plan.meta[d5], rpe_table=t5, n_tokens=nt5, n_windows=W5, patch_size=K, dilation=1, n_relay=0,
              n_heads=8, batch_size=len(sizes), rt_row0=0, depth=d5)
    want5 = ops.window_attention(qkv5, plan.meta[d5], t5, nt5, W5, K, 1, 0, 8, len(sizes), depth=d5, out_split=2, qkv_f16=True)
    mixed = ops.window_attention_multi([dict(probs[0]), dict(p5)])
    assert torch.equal(mixed[1][:nt5].view(torch.int16), want5[:nt5].view(torch.int16))
    used0 = probs[0]['n_tokens'] + (-(-probs[0]['n_tokens'] // K))
    assert torch.equal(mixed[0][:used0].view(torch.int16), single[0][:used0].view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize('c', [256, 128])
def test_row_segment_forms_equal_the_concatenated_input(c):
    """hfl_ln_qkv_fused_seg / hfl_linear_x3_seg (rows of the input / of the residual read from up to four arrays: the relay rows
    of the pyramid levels where the levels left them, models/hotformerloc_backbone.py:593-633) against the same launches on
    torch.cat of the parts: bitwise equal, for one to four parts, parts that end inside a row tile, many rows and few."""
    g = torch.Generator(device='cuda').manual_seed(21)
    gamma = torch.rand(c, device='cuda', generator=g) + 0.5
    beta = torch.randn(c, device='cuda', generator=g) * 0.1
    wq = torch.randn(3 * c, c, device='cuda', generator=g) * 0.08
    bq = torch.randn(3 * c, device='cuda', generator=g) * 0.1
    pack = ops.qkv_fused_pack(wq)
    wp = ops.split2(torch.randn(c, c, device='cuda', generator=g) * 0.08)
    bp = torch.randn(c, device='cuda', generator=g) * 0.1
    for sizes in ([1392, 292, 44], [5], [7, 1], [40000, 1392], [100, 28, 3, 129], [66775, 1392]):
        parts = [torch.randn(n, c, device='cuda', generator=g) for n in sizes]
        whole = torch.cat(parts, 0)
        a = ops.ln_qkv_fused(whole, gamma, beta, 1e-5, pack, bq, 0.36)
        b = ops.ln_qkv_fused(parts, gamma, beta, 1e-5, pack, bq, 0.36)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), ('ln_qkv', sizes)
        x2 = ops.split2(torch.randn(whole.shape[0], c, device='cuda', generator=g))
        ya = ops.linear_x3(x2, wp, bias=bp, residual=whole)
        yb = ops.linear_x3(x2, wp, bias=bp, residual=parts)
        assert torch.equal(ya, yb), ('linear', sizes)
    lib = __import__('hotformerloc_amd._native', fromlist=['x']).load()
    from hotformerloc_amd._native import RowSegments
    import ctypes
    bad = RowSegments()
    bad.n = 5
    assert lib.hfl_linear_x3_seg(ya.data_ptr(), x2.data_ptr(), wp.data_ptr(), None, ctypes.byref(bad), 4, c, c, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,octree_depth,sizes', [('wild-places', 7, [4096, 30, 2500, 4096, 4096]), ('oxford', 9, [4096, 1000]),
                                                    ('wild-places', 7, [4096] * 12)])
def test_attn_ws_equals_the_two_launches(cfg, octree_depth, sizes):
    """hfl_attn_ws_fwd (LayerNorm -> qkv -> window attention of a relay-token block in ONE launch with specialised GEMM /
    attention waves: C = 256, 16 heads, K = 48 tokens + the window's relay token, models/hotformerloc_backbone.py:197-216)
    against hfl_ln_qkv_fused over [token rows | relay rows] + the fp16 window kernel with the relay tokens (the path
    test_window_attention_matches_oracle pins to the oracle's materialised hat_window_mask + padded RPE).  Token rows: the
    same operations in the same order -> BITWISE equal (the two-launch kernel switched to the three-table RPE form the fused
    kernel always uses).  Relay rows: their one query per window is a query tile with one live column -> the same values (bar:
    one unit of the split2 output's low half).  With RPE also against the oracle's composition LayerNorm -> Linear -> attention
    on the reference's materialised hat_window_mask + padded RPE, <= 3e-5 of the largest value (the output is the proj GEMM's
    split-bf16 operand: 16 significant bits per value).  Every pyramid depth of the config, with and without RPE, ragged
    batches (partly padded last windows, windows that straddle clouds), more tiles than CUs (several units per workgroup, the
    left-over tiles split by head pairs), deterministic."""
    from hotformerloc_amd import _native
    lib = _native.load()
    clouds = [syn.unit_ball_cloud(700 + i, n) for i, n in enumerate(sizes)]
    params, ref, dev, oplan, plan = _plans(clouds, cfg, octree_depth)
    K = params.patch_size
    H, C = 16, 256
    g = torch.Generator(device='cuda').manual_seed(5)
    gamma = torch.rand(C, device='cuda', generator=g) + 0.5
    beta = torch.randn(C, device='cuda', generator=g) * 0.1
    w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.08
    b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
    qs = 16 ** -0.5 * 1.4426950408889634
    pack = ops.qkv_fused_pack(w)
    bnd = int(0.8 * K)
    try:
        lib.hfl_set_variant(b'window_rpe_form1_max_depth', 0)
        for depth in plan.pyramid_depths:
            nt, W = plan.n_tokens[depth], plan.n_windows[depth]
            assert ops.attn_ws_ok(nt, W, K, H, depth, C), (cfg, depth)
            x = torch.randn(nt + W, C, device='cuda', generator=g) * 1.3 + 0.2
            for with_rpe in (True, False):
                table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.3 if with_rpe else None
                qkv = ops.ln_qkv_fused(x, gamma, beta, 1e-5, pack, b, qs)
                two = ops.window_attention(qkv, plan.meta[depth], table, nt, W, K, 1, 1, H, plan.B, rt_row0=nt, depth=depth,
                                           out_split=2, qkv_f16=True)
                one = ops.attn_ws(x[:nt], gamma, beta, 1e-5, pack, b, qs, qkv[nt:], plan.meta[depth], table, nt, W, K, H,
                                  plan.B, depth)
                nbad = (one[:nt].view(torch.int16) != two[:nt].view(torch.int16)).any(dim=1).sum().item()
                assert nbad == 0, (cfg, depth, with_rpe, nbad)

                def val(t):
                    v = t[nt:nt + W].float().view(W, C // 32, 2, 32)
                    return (v[:, :, 0] + v[:, :, 1]).reshape(W, C)
                a, r = val(one), val(two)
                assert torch.isfinite(a).all()
                err = (a - r).abs().max().item()
                assert err <= 1.2e-5 * max(r.abs().max().item(), 1.0), (cfg, depth, with_rpe, err)
                again = ops.attn_ws(x[:nt], gamma, beta, 1e-5, pack, b, qs, qkv[nt:], plan.meta[depth], table, nt, W, K, H,
                                    plan.B, depth)
                assert torch.equal(one.view(torch.int16), again.view(torch.int16))
                if with_rpe:
                    # ... and against the ORACLE composition in fp64 -> fp32: LayerNorm -> Linear -> scaled-dot-product attention
                    # on the reference's materialised windows, hat_window_mask and zero-padded RPE
                    # (models/hotformerloc_backbone.py:197-214, models/octformer_backbone.py:52-93)
                    xd = x.cpu().double()
                    ln = torch.nn.functional.layer_norm(xd, (C,), gamma.cpu().double(), beta.cpu().double(), 1e-5)
                    q64 = (ln @ w.cpu().double().t() + b.cpu().double()).float()
                    xw = torch.cat([q64[nt:].unsqueeze(1), oplan.to_windows(q64[:nt], depth, False)], 1)
                    q_, k_, v_ = xw.reshape(-1, K + 1, 3, H, 16).permute(2, 0, 3, 1, 4)
                    rpe = torch.nn.functional.pad(hotformer_ref.rpe_bias(table.cpu(), oplan.rel_pos[depth], K, 1), (1, 0, 1, 0))
                    want = hotformer_ref._sdpa(q_, k_, v_, oplan.hat_mask[depth].unsqueeze(1) + rpe, 0.25)
                    want = want.transpose(1, 2).reshape(-1, K + 1, C)
                    got = one.float().cpu().view(nt + W, C // 32, 2, 32)
                    got = (got[:, :, 0] + got[:, :, 1]).reshape(nt + W, C)
                    e_tok = (got[:nt] - oplan.from_windows(want[:, 1:], depth, False)).abs().max().item()
                    real_w = -(-nt // K)
                    e_rt = (got[nt:nt + real_w] - want[:real_w, 0]).abs().max().item()
                    print('attn_ws vs oracle composition', cfg, depth, 'token rows %.2e relay rows %.2e' % (e_tok, e_rt))
                    assert max(e_tok, e_rt) <= 3e-5 * max(want.abs().max().item(), 1.0), (cfg, depth, e_tok, e_rt)
    finally:
        lib.hfl_set_variant(b'window_rpe_form1_max_depth', 4)
    assert not ops.attn_ws_ok(1000, 21, 48, 8, 4, 128)            # C = 128 blocks have no relay tokens: hfl_attn_fused_fwd
    assert not ops.attn_ws_ok(1000, 16, 64, 16, 4, 256)           # K = 64 (CS-Wild-Places): the two launches


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,octree_depth,sizes', [('wild-places', 7, [4096, 30, 2500, 4096]), ('cs-wild-places', 7, [6000, 4096]),
                                                    ('oxford', 9, [4096, 1000])])
def test_attn_fused_equals_the_two_launches(cfg, octree_depth, sizes):
    """hfl_attn_fused_fwd (LayerNorm -> qkv -> window attention in ONE launch, q / k / v never in HBM; built for the OctFormer
    stage: C = 128, K = 48, no relay tokens) against (a) hfl_ln_qkv_fused + the fp16 window kernel, whose arithmetic it
    repeats operation for operation: BITWISE equal, dilation 1 and 4, with and without RPE, ragged batches whose last tile
    and windows are partly padding, at the attention depths of the three configs (5, 5, 7).  The two-launch path is the one
    the oracle tests above pin (test_ln_qkv_fused_*, test_window_attention_matches_oracle): equal bits carry that parity over."""
    clouds = [syn.unit_ball_cloud(900 + i, n) for i, n in enumerate(sizes)]
    params, ref, dev, oplan, plan = _plans(clouds, cfg, octree_depth)
    K = params.patch_size
    depth = plan.stage_depths[0] if hasattr(plan, 'stage_depths') else max(plan.n_tokens.keys())
    H, C = 8, 128
    nt = plan.n_tokens[depth]
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(nt, C, device='cuda', generator=g) * 1.3 + 0.2
    gamma = torch.rand(C, device='cuda', generator=g) + 0.5
    beta = torch.randn(C, device='cuda', generator=g) * 0.1
    w = torch.randn(3 * C, C, device='cuda', generator=g) * 0.08
    b = torch.randn(3 * C, device='cuda', generator=g) * 0.1
    qs = 16 ** -0.5 * 1.4426950408889634
    pack = ops.qkv_fused_pack(w)
    for dil in (1, 4):
        W = -(-nt // (K * dil)) * dil
        bnd = int(0.8 * K * dil ** 0.5)
        for with_rpe in (True, False):
            table = torch.randn(3 * (2 * bnd + 1), H, device='cuda', generator=g) * 0.3 if with_rpe else None
            if K != 48:                                     # (CS-Wild-Places windows hold 64 tokens: not a shape it is built for)
                assert not ops.attn_fused_ok(nt, W, K, dil, 0, H, depth, C, with_rpe)
                continue
            assert ops.attn_fused_ok(nt, W, K, dil, 0, H, depth, C, with_rpe), (cfg, depth, dil, with_rpe)
            qkv = ops.ln_qkv_fused(x, gamma, beta, 1e-5, pack, b, qs)
            two = ops.window_attention(qkv, plan.meta[depth], table, nt, W, K, dil, 0, H, plan.B, rt_row0=nt, depth=depth,
                                       out_split=2, qkv_f16=True)
            one = ops.attn_fused(x, gamma, beta, 1e-5, pack, b, qs, plan.meta[depth], table, nt, W, K, dil, H, plan.B, depth)
            assert torch.equal(one[:nt].view(torch.int16), two[:nt].view(torch.int16)), (cfg, dil, with_rpe)
            assert torch.equal(one, ops.attn_fused(x, gamma, beta, 1e-5, pack, b, qs, plan.meta[depth], table, nt, W, K, dil,
                                                   H, plan.B, depth))
