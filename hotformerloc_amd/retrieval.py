"""Retrieval metric of the reference's evaluation (`eval/pnv_evaluate.py:199-315`) on the GPU: exact flat-L2
top-25 search of every query descriptor in a database set -- one (Q, D) x (D, N) GEMM plus a top-k instead of a
FAISS index / sklearn KDTree -- then recall@1..25, top-1 % recall and mean reciprocal rank, with the reference's
`get_recall` signature and return value."""

import numpy as np
import torch

from ._native import NativeLibraryError


def flat_l2_topk(database: torch.Tensor, queries: torch.Tensor, k: int):
    """(N, D), (Q, D) on the GPU -> squared-L2 distances and indices (Q, k), nearest first (ties: lower index)."""
    if database.device.type != 'cuda' or queries.device.type != 'cuda':
        raise NativeLibraryError('flat_l2_topk runs on the GPU only (no CPU fallback)')
    db, q = database.double(), queries.double()           # (Q, N) of a few thousand: f64 keeps the ranking exact
    d2 = (q * q).sum(1, keepdim=True) + (db * db).sum(1)[None, :] - 2.0 * (q @ db.t())
    order = torch.argsort(d2, dim=1, stable=True)[:, :k]
    return d2.gather(1, order), order


def get_recall(m, n, database_vectors, query_vectors, query_sets, database_sets=None, log=False,
               model_name: str = 'model', device='cuda'):
    """`eval/pnv_evaluate.py:226-315` (the per-query false-positive logging of `log=True` is not reproduced)."""
    if log:
        raise NotImplementedError('log=True writes the reference\'s debug text files; not part of the metric')
    db = torch.as_tensor(np.asarray(database_vectors[m]), dtype=torch.float32, device=device)
    qs = torch.as_tensor(np.asarray(query_vectors[n]), dtype=torch.float32, device=device)
    num_neighbors = 25
    k = min(num_neighbors, db.shape[0])
    _, idx = flat_l2_topk(db, qs, k)
    n_q, n_db = qs.shape[0], db.shape[0]
    truth = torch.zeros((n_q, n_db), dtype=torch.bool)
    has_truth = torch.zeros(n_q, dtype=torch.bool)
    for i in range(n_q):
        tn = query_sets[n][i][m]
        if len(tn) > 0:
            truth[i, torch.as_tensor(list(tn), dtype=torch.long)] = True
            has_truth[i] = True
    truth, has_truth = truth.to(device), has_truth.to(device)
    hits = truth.gather(1, idx) & has_truth[:, None]                       # (Q, k): is the j-th result a true neighbour
    evaluated = int(has_truth.sum().item())
    any_hit = hits.any(1)
    first = torch.where(any_hit, hits.float().argmax(1), torch.full((n_q,), -1, device=device, dtype=torch.long))
    recall = torch.zeros(num_neighbors, dtype=torch.float64, device=device)
    recall.index_add_(0, first[any_hit], torch.ones(int(any_hit.sum().item()), dtype=torch.float64, device=device))
    threshold = max(int(round(n_db / 100.0)), 1)
    one_percent = int(hits[:, :min(threshold, k)].any(1).sum().item())
    recall = (torch.cumsum(recall, 0) / float(evaluated) * 100).cpu().numpy()
    one_percent_recall = (one_percent / float(evaluated)) * 100
    mrr = float((1.0 / (first[any_hit].double() + 1.0)).mean().item() * 100)
    return recall, one_percent_recall, mrr
