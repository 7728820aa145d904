"""GPU TruncatedSmoothAP (hotformerloc_amd/losses.py, hfl_smoothap_rows) against the golden vectors of the
reference's own loss and, at the reference's training batch size, against the CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hotformerloc_amd.losses import TruncatedSmoothAP      # noqa: E402
from oracle import loss_ref                                # noqa: E402
from oracle.gen_golden_loss import make_case               # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', ['b64', 'b48_few_pos', 'b96_p2'])
def test_loss_matches_reference_golden(golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'loss_smoothap.npz'))
    seed, batch, dim, group, drop, ppq = [int(v) for v in g[case + '.cfg']]
    e, pos, neg = make_case(seed, batch, dim, group, drop)
    emb = torch.from_numpy(e).cuda().requires_grad_()
    loss, stats = TruncatedSmoothAP(tau1=0.01, positives_per_query=ppq)(emb, torch.from_numpy(pos), torch.from_numpy(neg))
    loss.backward()
    assert abs(loss.item() - float(g[case + '.loss'])) < 2e-6
    gref = g[case + '.grad']
    assert np.abs(emb.grad.cpu().numpy() - gref).max() <= 2e-5 * max(np.abs(gref).max(), 1e-6) + 1e-7
    got = [stats['positives_per_query'], stats['ap'], stats['avg_embedding_norm']]
    assert np.allclose(got, g[case + '.stats'][[0, 3, 4]], atol=2e-6)
    if drop == 0:          # rows without positives pick an arbitrary "best positive" in the reference
        assert np.allclose([stats['best_positive_ranking'], stats['recall'][1]], g[case + '.stats'][[1, 2]], atol=1e-6)


def test_loss_at_training_batch_size_matches_oracle():
    """batch_size = 2048, positives_per_query = 4, tau1 = 0.01 (config/config_wild-places.txt:7,25,26)."""
    # groups of 5 -> exactly 4 positives per query: the selected set cannot flip on a near-tie between the CPU
    # and GPU similarity matrices (a flip changes the gradient of that row discretely, not the loss)
    e, pos, neg = make_case(21, 2048, 256, 5, 17)
    emb = torch.from_numpy(e).cuda().requires_grad_()
    loss, stats = TruncatedSmoothAP(tau1=0.01, positives_per_query=4)(emb, torch.from_numpy(pos), torch.from_numpy(neg))
    loss.backward()
    ref = torch.from_numpy(e).requires_grad_()
    want, wstats = loss_ref.truncated_smooth_ap(ref, torch.from_numpy(pos), torch.from_numpy(neg), 0.01, 4)
    want.backward()
    assert abs(loss.item() - want.item()) < 5e-6
    gref = ref.grad.numpy()
    # tau1 = 0.01 multiplies every rounding difference of E E^T by 100 inside the sigmoid: 3e-4 of the largest
    # gradient entry is the observed fp32 noise floor between the GPU GEMM and the CPU one (bar: 1e-3)
    err = np.abs(emb.grad.cpu().numpy() - gref).max() / np.abs(gref).max()
    print('B=2048 loss', loss.item(), want.item(), 'grad err / max', err)
    assert err <= 3e-4
    assert abs(stats['ap'] - wstats['ap']) < 5e-6


def test_loss_rejects_cpu_tensors():
    from hotformerloc_amd._native import NativeLibraryError
    with pytest.raises(NativeLibraryError):
        TruncatedSmoothAP()(torch.zeros(4, 8), torch.zeros(4, 4, dtype=torch.bool), torch.zeros(4, 4, dtype=torch.bool))
