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


def test_multistaged_step_on_the_encoder_matches_oracle_chain():
    """SURVEY 8f rank 1 end to end on one GPU: two minibatches through the HIP encoder, GPU TruncatedSmoothAP,
    stage-3 back-propagation -- against (a) direct autograd through both minibatches on the GPU and (b) the
    CPU oracle chain (oracle encoder + loss oracle, torch autograd).  drop_path = 0 (RNG parity is impossible)."""
    from hotformerloc_amd import build_batch_octree, load_config, model_factory
    from hotformerloc_amd import synthetic as syn
    from hotformerloc_amd.training import multistaged_training_step
    from oracle import hotformer_ref
    from oracle.testing import oracle_octree, synthetic_state_dict
    params, depth = load_config('wild-places')
    params.drop_path = 0.0
    clouds = [syn.cylindrical(syn.unit_ball_cloud(3100 + i, 700 + 100 * i)) for i in range(4)]      # sized for the CPU oracle's
    parts = [clouds[:2], clouds[2:]]                                                                  # forward + backward
    lab = torch.arange(4) // 2
    pos = (lab[:, None] == lab[None, :]) & ~torch.eye(4, dtype=torch.bool)
    neg = lab[:, None] != lab[None, :]
    loss_fn = TruncatedSmoothAP(tau1=0.01, positives_per_query=1)

    def fresh():
        m = model_factory(params)
        syn.fill_synthetic_weights(m, 'stress')
        return m.cuda()

    model = fresh()
    mbs = [{'octree': build_batch_octree(p, depth, 2, 'cuda')} for p in parts]
    stats = multistaged_training_step(model, mbs, pos, neg, loss_fn)
    # (a) direct autograd on the GPU
    direct = fresh().train()
    emb = torch.cat([direct({'octree': build_batch_octree(p, depth, 2, 'cuda')})['global'] for p in parts], 0)
    loss, _ = loss_fn(emb, pos, neg)
    loss.backward()
    assert abs(stats['loss'] - loss.item()) < 1e-5
    for (n, p), q in zip(model.named_parameters(), direct.parameters()):
        d = (p.grad - q.grad).norm().item()
        assert d <= 2e-4 * max(q.grad.norm().item(), 1e-9) + 1e-9, (n, d, q.grad.norm().item())
    # (b) the CPU oracle chain
    sd = {k: v.clone().requires_grad_() for k, v in synthetic_state_dict(params, 'stress').items()}
    emb_ref = torch.cat([hotformer_ref.forward_with_grad(sd, params, oracle_octree(p, depth)) for p in parts], 0)
    want, _ = loss_ref.truncated_smooth_ap(emb_ref, pos, neg, 0.01, 1)
    want.backward()
    assert abs(stats['loss'] - want.item()) < 1e-4
    worst = 0.0
    for n, p in model.named_parameters():
        gref = sd[n].grad
        err = (p.grad.cpu() - gref).norm().item() / max(gref.norm().item(), 1e-12)
        worst = max(worst, err if gref.norm().item() > 1e-9 else 0.0)
        assert err < 1e-3 or gref.norm().item() < 1e-9, (n, err, gref.norm().item())      # the forward's bar (north star: 1e-3)
    print('multistaged step: loss', stats['loss'], want.item(), 'worst param-grad rel-L2 vs oracle', worst)
