"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's listwise loss,
`models/losses/truncated_smoothap.py:10-99` with `models/losses/loss_utils.py:40-63` (temperature
sigmoid with the exponent clamped to [-50, 50], cosine affinity = E E^T), written for torch autograd so
that d loss / d embeddings comes out of the same graph the reference differentiates.

Pinned against the reference module itself: `oracle/gen_golden_loss.py` imports
`/root/reference/models/losses/truncated_smoothap.py` in the build container and stores inputs,
loss, statistics and the embedding gradient in `tests/golden/loss_smoothap.npz`
(`tests/test_oracle_loss.py`)."""

import torch


def temperature_sigmoid(x: torch.Tensor, temp: float) -> torch.Tensor:
    """loss_utils.py:40-48"""
    e = torch.clamp(-x / temp, min=-50, max=50)
    return 1.0 / (1.0 + torch.exp(e))


def truncated_smooth_ap(embeddings: torch.Tensor, positives_mask: torch.Tensor, negatives_mask: torch.Tensor,
                        tau1: float = 0.01, positives_per_query: int = 4):
    """Returns (loss, stats) as `TruncatedSmoothAP.__call__` (truncated_smoothap.py:22-99), cosine similarity."""
    s = embeddings @ embeddings.t()                                              # :33
    sp = s.detach().clone()
    sp.masked_fill_(~positives_mask, float('-inf'))                              # :36-37
    idx = torch.topk(sp, k=positives_per_query, dim=1, largest=True, sorted=True)[1]      # :39
    n_pos = positives_mask.sum(1)
    s_diff = s.unsqueeze(1) - s.gather(1, idx).unsqueeze(2)                      # (B,P,B)  :46
    sg = temperature_sigmoid(s_diff, tau1)
    pos = sg * positives_mask.unsqueeze(1)                                       # :51-52
    pos = pos * torch.ones_like(pos).scatter(2, idx.unsqueeze(2), 0.)            # :55-56
    r_p = pos.sum(2) + 1.0                                                       # :59
    r_omega = r_p + (sg * negatives_mask.unsqueeze(1)).sum(2)                    # :64-66
    r = r_p / r_omega
    hard = torch.logical_and((s_diff.detach() > 0)[:, 0], negatives_mask).sum(1)  # :76-78
    valid = torch.gather(positives_mask, 1, idx)                                 # :85
    n_valid = valid.sum(1)
    q = n_valid > 0                                                              # :90
    ap = ((r * valid)[q].sum(1) / n_valid[q]).mean()                             # :93
    loss = 1.0 - ap
    stats = {'positives_per_query': n_pos.float().mean().item(),
             'best_positive_ranking': hard.float().mean().item(),
             'recall': {1: (hard <= 1).float().mean().item()},
             'loss': loss.item(), 'ap': ap.item(),
             'avg_embedding_norm': embeddings.norm(dim=1).mean().item()}
    return loss, stats
