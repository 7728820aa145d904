"""Retrieval-metric oracle against goldens produced by the reference's own get_recall (KDTree branch)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import retrieval_ref                                # noqa: E402
from oracle.gen_golden_retrieval import make_sets              # noqa: E402


def _pairs(g, name):
    n_sets = int(g[name + '.cfg'][1])
    return [(m, n) for m in range(n_sets) for n in range(n_sets) if m != n]


@pytest.mark.parametrize('name', ['small', 'wide'])
def test_retrieval_oracle_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'retrieval.npz'))
    seed, n_sets, per_set, dim, places = [int(v) for v in g[name + '.cfg']]
    vecs, qsets = make_sets(seed, n_sets, per_set, dim, places)
    for m, n in _pairs(g, name):
        recall, opr, mrr = retrieval_ref.get_recall(m, n, vecs, vecs, qsets)
        assert np.allclose(recall, g['%s.%d.%d.recall' % (name, m, n)], atol=1e-9)
        assert np.allclose([opr, mrr], g['%s.%d.%d.opr_mrr' % (name, m, n)], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['small', 'wide'])
def test_gpu_retrieval_matches_reference_golden(golden_dir, name):
    from hotformerloc_amd.retrieval import get_recall
    g = np.load(os.path.join(golden_dir, 'retrieval.npz'))
    seed, n_sets, per_set, dim, places = [int(v) for v in g[name + '.cfg']]
    vecs, qsets = make_sets(seed, n_sets, per_set, dim, places)
    for m, n in _pairs(g, name):
        recall, opr, mrr = get_recall(m, n, vecs, vecs, qsets, None)
        assert np.allclose(recall, g['%s.%d.%d.recall' % (name, m, n)], atol=1e-9)
        assert np.allclose([opr, mrr], g['%s.%d.%d.opr_mrr' % (name, m, n)], atol=1e-9)
