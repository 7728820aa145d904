"""Listwise loss of the reference's training step on the GPU: `TruncatedSmoothAP`
(`models/losses/truncated_smoothap.py:10-99`, built by `models/losses/loss.py:17-19` from
`tau1`, `similarity`, `positives_per_query` of the training config).  Same constructor, same call
signature `(embeddings, positives_mask, negatives_mask) -> (loss, stats)`, same `stats` keys.

The (B, P, B) ranking algebra and its gradient run in one HIP kernel per call (`hfl_smoothap_rows`);
the affinity matrix, the top-k selection of positives and dE = (dS + dS^T) E are dense torch ops."""

import numpy as np
import torch

from . import _native
from ._native import check
from . import ops


class _SmoothAPRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sim, pos_u8, neg_u8, idx, tau):
        b = sim.shape[0]
        ap = torch.empty(b, dtype=torch.float32, device=sim.device)
        dap = torch.empty((b, b), dtype=torch.float32, device=sim.device)
        check(_native.load().hfl_smoothap_rows(ap.data_ptr(), dap.data_ptr(), sim.data_ptr(), pos_u8.data_ptr(),
                                               neg_u8.data_ptr(), idx.data_ptr(), b, idx.shape[1], float(tau),
                                               ops._stream()), 'hfl_smoothap_rows')
        ctx.save_for_backward(dap)
        return ap

    @staticmethod
    def backward(ctx, grad_ap):
        (dap,) = ctx.saved_tensors
        return dap * grad_ap[:, None], None, None, None, None


class TruncatedSmoothAP:
    def __init__(self, tau1: float = 0.01, similarity: str = 'cosine', positives_per_query: int = 4):
        if similarity != 'cosine':
            raise NotImplementedError("similarity=%r: every shipped training config uses 'cosine'" % similarity)
        self.tau1 = tau1
        self.similarity = similarity
        self.positives_per_query = positives_per_query

    def __call__(self, embeddings, positives_mask, negatives_mask):
        device = embeddings.device
        if device.type != 'cuda':
            raise _native.NativeLibraryError('TruncatedSmoothAP runs on the GPU only (no CPU fallback)')
        positives_mask = positives_mask.to(device)
        negatives_mask = negatives_mask.to(device)
        emb = embeddings.float()
        s_qz = emb @ emb.t()                                                       # compute_aff, cosine
        s_pos = s_qz.detach().clone()
        s_pos.masked_fill_(torch.logical_not(positives_mask), -np.inf)
        idx = torch.topk(s_pos, k=self.positives_per_query, dim=1, largest=True, sorted=True)[1]
        n_positives = positives_mask.sum(dim=1)
        valid = torch.gather(positives_mask, 1, idx)
        n_valid = valid.sum(dim=1)
        valid_q = n_valid > 0
        ap_rows = _SmoothAPRows.apply(s_qz.contiguous(), positives_mask.to(torch.uint8).contiguous(),
                                      negatives_mask.to(torch.uint8).contiguous(), idx.contiguous(), self.tau1)
        ap = ap_rows[valid_q].mean()
        loss = 1. - ap
        with torch.no_grad():                                                      # truncated_smoothap.py:71-80
            best = s_qz.detach().gather(1, idx[:, :1])
            hard_ranking = torch.logical_and(s_qz.detach() > best, negatives_mask).sum(dim=1)
            stats = {'positives_per_query': n_positives.float().mean(dim=0).item(),
                     'best_positive_ranking': hard_ranking.float().mean(dim=0).item(),
                     'recall': {1: (hard_ranking <= 1).float().mean(dim=0).item()},
                     'loss': loss.item(), 'ap': ap.item(),
                     'avg_embedding_norm': embeddings.norm(dim=1).mean().item()}
        return loss, stats
