"""The loss oracle (oracle/loss_ref.py) against golden vectors produced by the reference's own
TruncatedSmoothAP (oracle/gen_golden_loss.py, models/losses/truncated_smoothap.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import loss_ref                       # noqa: E402
from oracle.gen_golden_loss import make_case      # noqa: E402

CASES = ['b64', 'b48_few_pos', 'b96_p2']


@pytest.mark.parametrize('case', CASES)
def test_loss_oracle_matches_reference_golden(golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'loss_smoothap.npz'))
    seed, batch, dim, group, drop, ppq = [int(v) for v in g[case + '.cfg']]
    e, pos, neg = make_case(seed, batch, dim, group, drop)
    emb = torch.from_numpy(e).requires_grad_()
    loss, stats = loss_ref.truncated_smooth_ap(emb, torch.from_numpy(pos), torch.from_numpy(neg), 0.01, ppq)
    loss.backward()
    assert abs(loss.item() - float(g[case + '.loss'])) < 1e-6
    assert np.abs(emb.grad.numpy() - g[case + '.grad']).max() <= 1e-6 * max(1.0, np.abs(g[case + '.grad']).max())
    got = [stats['positives_per_query'], stats['best_positive_ranking'], stats['recall'][1], stats['ap'],
           stats['avg_embedding_norm']]
    assert np.allclose(got, g[case + '.stats'], atol=1e-6)
