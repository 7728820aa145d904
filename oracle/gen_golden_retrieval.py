"""Build-container only: run the reference's `get_recall` (`/root/reference/eval/pnv_evaluate.py:226-315`, KDTree
branch -- FAISS is not installed) on closed-form descriptor sets -> tests/golden/retrieval.npz.
The module itself cannot be imported here (it pulls the whole evaluation stack), so the three functions it needs are
executed from its source text in an isolated namespace; nothing of that text is stored in the repo."""
import ast
import os
import sys

import numpy as np
from sklearn.neighbors import KDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hotformerloc_amd import synthetic as syn    # noqa: E402


def make_sets(seed: int, n_sets: int, per_set: int, dim: int, places: int, jitter: float = 2.2):
    """`n_sets` traversals of the same `places` places (+ jitter): descriptors (per_set, dim) per set and the
    reference's query-set dictionaries {i: {m: [indices of true neighbours in set m]}} (some queries have none)."""
    base = syn.hash_uniform(seed, places * dim).reshape(places, dim) - 0.5
    vecs, place_of = [], []
    for s in range(n_sets):
        pl = (np.arange(per_set) * 7 + 3 * s) % places
        v = base[pl] + jitter * (syn.hash_uniform(seed + 10 + s, per_set * dim).reshape(per_set, dim) - 0.5)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        vecs.append(v.astype(np.float32))
        place_of.append(pl)
    query_sets = []
    for n in range(n_sets):
        qd = {}
        for i in range(per_set):
            qd[i] = {m: [int(j) for j in np.nonzero(np.abs(place_of[m] - place_of[n][i]) <= 1)[0]]
                        if (i + n) % 11 else [] for m in range(n_sets)}
        query_sets.append(qd)
    return vecs, query_sets


def reference_get_recall():
    src = open('/root/reference/eval/pnv_evaluate.py').read()
    tree = ast.parse(src)
    keep = [node for node in tree.body if isinstance(node, ast.FunctionDef)
            and node.name in ('_build_index', '_index_search', 'get_recall')]
    ns = {'np': np, 'KDTree': KDTree, 'HAS_FAISS': False}
    exec(compile(ast.Module(body=keep, type_ignores=[]), 'pnv_evaluate_subset', 'exec'), ns)
    return ns['get_recall']


def main():
    get_recall = reference_get_recall()
    out = {}
    for name, (seed, n_sets, per_set, dim, places) in {'small': (5, 3, 60, 32, 40), 'wide': (6, 2, 400, 256, 150)}.items():
        vecs, qsets = make_sets(seed, n_sets, per_set, dim, places)
        out[name + '.cfg'] = np.array([seed, n_sets, per_set, dim, places])
        for m in range(n_sets):
            for n in range(n_sets):
                if m == n:
                    continue
                recall, opr, mrr = get_recall(m, n, vecs, vecs, qsets, None)
                out['%s.%d.%d.recall' % (name, m, n)] = np.asarray(recall, dtype=np.float64)
                out['%s.%d.%d.opr_mrr' % (name, m, n)] = np.array([opr, mrr], dtype=np.float64)
                print(name, m, n, recall[:3], opr, mrr)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'retrieval.npz'), **out)


if __name__ == '__main__':
    main()
