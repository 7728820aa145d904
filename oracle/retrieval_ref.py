"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's retrieval metric, `get_recall`
(`eval/pnv_evaluate.py:226-315`): flat L2 top-25 search of every query descriptor in a database set
(`_build_index` / `_index_search`, :199-223 -- FAISS or an sklearn KDTree, both exact L2), then
recall@1..25 (first true neighbour among the 25), top-1% recall and mean reciprocal rank.

Pinned by `oracle/gen_golden_retrieval.py`, which executes the reference's own function (KDTree branch)
in the build container -> `tests/golden/retrieval.npz` (`tests/test_oracle_retrieval.py`)."""

import numpy as np


def get_recall(m, n, database_vectors, query_vectors, query_sets, database_sets=None, num_neighbors: int = 25):
    db = np.asarray(database_vectors[m], dtype=np.float32)
    qs = np.asarray(query_vectors[n], dtype=np.float32)
    k = min(num_neighbors, len(db))
    d2 = ((qs[:, None, :].astype(np.float64) - db[None, :, :].astype(np.float64)) ** 2).sum(-1)
    idx = np.argsort(d2, axis=1, kind='stable')[:, :k]                        # :236 exact L2 ranking
    recall = [0] * num_neighbors
    recall_idx = []
    one_percent = 0
    threshold = max(int(round(len(db) / 100.0)), 1)                           # :233
    evaluated = 0
    for i in range(len(qs)):
        true_nb = query_sets[n][i][m]
        if len(true_nb) == 0:                                                 # :243-244
            continue
        evaluated += 1
        for j in range(idx.shape[1]):                                         # :300-304
            if idx[i][j] in true_nb:
                recall[j] += 1
                recall_idx.append(j + 1)
                break
        if len(set(idx[i][0:threshold]).intersection(set(true_nb))) > 0:      # :306-307
            one_percent += 1
    one_percent_recall = (one_percent / float(evaluated)) * 100
    recall = (np.cumsum(recall) / float(evaluated)) * 100
    mrr = np.mean(1 / np.array(recall_idx)) * 100
    return recall, one_percent_recall, mrr
