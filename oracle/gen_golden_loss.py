"""Build-container only: golden vectors of the reference's TruncatedSmoothAP
(`/root/reference/models/losses/truncated_smoothap.py`) -> tests/golden/loss_smoothap.npz.
Inputs are closed-form (hash_uniform), so the file only pins outputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hotformerloc_amd import synthetic as syn    # noqa: E402


def make_case(seed: int, batch: int, dim: int, group: int, drop_rows: int):
    """unit-norm embeddings; positives = same group of `group` consecutive items (not self), negatives = the
    items further than one group away; the first `drop_rows` queries have no positive at all."""
    e = syn.hash_uniform(seed, batch * dim).reshape(batch, dim).astype(np.float64) - 0.5
    lab = np.arange(batch) // group
    e += 0.6 * (syn.hash_uniform(seed + 1, (batch // group + 1) * dim).reshape(-1, dim) - 0.5)[lab]
    e = (e / np.linalg.norm(e, axis=1, keepdims=True)).astype(np.float32)
    pos = (lab[:, None] == lab[None, :]) & ~np.eye(batch, dtype=bool)
    neg = np.abs(lab[:, None] - lab[None, :]) > 1
    pos[:drop_rows] = False
    return e, pos, neg


def main():
    sys.path.insert(0, '/root/reference')
    if not hasattr(np, 'NINF'):
        np.NINF = -np.inf          # the reference targets numpy 1.x (truncated_smoothap.py:37); same value
    saved = sys.modules.pop('datasets', None)
    from models.losses.truncated_smoothap import TruncatedSmoothAP       # the reference itself
    if saved is not None:
        sys.modules['datasets'] = saved
    out = {}
    for name, (seed, batch, dim, group, drop, ppq) in {
            'b64': (11, 64, 256, 4, 0, 4), 'b48_few_pos': (12, 48, 256, 3, 5, 4), 'b96_p2': (13, 96, 128, 6, 2, 2)}.items():
        e, pos, neg = make_case(seed, batch, dim, group, drop)
        emb = torch.from_numpy(e).requires_grad_()
        loss, stats = TruncatedSmoothAP(tau1=0.01, positives_per_query=ppq)(emb, torch.from_numpy(pos), torch.from_numpy(neg))
        loss.backward()
        out[name + '.cfg'] = np.array([seed, batch, dim, group, drop, ppq])
        out[name + '.loss'] = np.float32(loss.item())
        out[name + '.grad'] = emb.grad.numpy()
        out[name + '.stats'] = np.array([stats['positives_per_query'], stats['best_positive_ranking'],
                                         stats['recall'][1], stats['ap'], stats['avg_embedding_norm']], dtype=np.float64)
        print(name, loss.item(), stats)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'loss_smoothap.npz'), **out)


if __name__ == '__main__':
    main()
