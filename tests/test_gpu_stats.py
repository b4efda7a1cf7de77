"""stats.hip: the one-launch statistics of the actor-critic update against the torch expressions of the reference
(agent/dreamer_utils.py:934-1029, agent/dreamer.py:392-438)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [2, 17, 16384, 48000, 300001])
def test_quantile_ema_matches_torch_quantile(n):
    from genrl_amd import ops
    g = torch.Generator(device='cuda').manual_seed(n)
    x = torch.randn(n, device='cuda', generator=g) * 3 + 1
    if n > 100:
        x[::7] = x[3]                      # ties
        x[5] = -0.0; x[6] = 0.0
    ema = torch.tensor([0.3, 2.0], device='cuda')
    ref_q = torch.quantile(x, torch.tensor([0.05, 0.95], device='cuda'))
    ref_ema = 0.01 * ref_q + 0.99 * ema
    out = ops.quantile_ema(x, ema, 0.01)
    # the same order statistics and interpolation weights; the final lerp may differ in its last bit (fma contraction)
    assert torch.allclose(out[2:], ref_q, rtol=3e-7, atol=0), (out[2:], ref_q)
    assert torch.allclose(ema, ref_ema, rtol=1e-6, atol=0)
    assert torch.equal(out[0], ema[0]) and torch.equal(out[1], torch.clip(ema[1] - ema[0], min=1.0))


def test_moments_wmean_entropy_actor_objective():
    from genrl_amd import ops
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(17, 1024, 1, device='cuda', generator=g) * 2 + 0.5
    m = ops.moments(x)
    ref = torch.stack([x.mean(), x.std(), x.abs().mean(), (x * x).mean()])
    assert torch.allclose(m, ref, rtol=2e-6)
    # weighted mean with gradient
    a = torch.randn(16, 1024, device='cuda', generator=g, requires_grad=True)
    w = torch.rand(16, 1024, device='cuda', generator=g)
    l = ops.wmean(a, w, -1.0); l.backward()
    a2 = a.detach().clone().requires_grad_(True)
    l2 = -(a2 * w).mean(); l2.backward()
    assert torch.allclose(l, l2, rtol=2e-6) and torch.allclose(a.grad, a2.grad, rtol=1e-6)
    assert torch.allclose(ops.wmean(a.detach(), None, 1.0), a.detach().mean(), rtol=2e-6, atol=1e-7)
    # policy entropy metric
    raw = torch.randn(15, 1024, 20, device='cuda', generator=g)
    std = 0.9 * torch.sigmoid(raw[..., 10:] + 2.0) + 0.1
    ref = (0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std)).sum(-1).mean()
    assert torch.allclose(ops.normal_entropy_mean(raw, 0.1, 1.0), ref, rtol=2e-6)
    # actor objective: -(weight[:-2] * ((target - offset) / scale)[1:]).mean() and the normed statistics
    H, N = 16, 1024
    t = (torch.randn(H, N, 1, device='cuda', generator=g) * 5).requires_grad_(True)
    wt = torch.rand(H + 1, N, 1, device='cuda', generator=g)
    os_ = torch.tensor([0.7, 3.1, 0.0, 0.0], device='cuda')
    loss, st = ops.actor_objective(t, wt[:-2], os_)
    loss.backward()
    t2 = t.detach().clone().requires_grad_(True)
    normed = (t2 - os_[0]) / os_[1]
    l2 = -(wt[:-2] * normed[1:]).mean(); l2.backward()
    assert torch.allclose(loss, l2, rtol=3e-6) and torch.allclose(t.grad, t2.grad, rtol=1e-6)
    assert torch.allclose(st[0], normed.mean(), rtol=1e-5, atol=1e-7) and torch.allclose(st[1], normed.std(), rtol=1e-5)
    loss_u, _ = ops.actor_objective(t.detach(), None, os_)             # unit weight
    assert torch.allclose(loss_u, -normed[1:].mean().detach(), rtol=3e-6)


@pytest.mark.parametrize('momentum', [0.9, 0.0, 1.0])
def test_stream_norm_ema_matches_reference_recurrence(momentum):
    """StreamNorm with momentum != 1 (plan2explore.yaml's reward_norm; agent/dreamer_utils.py:966-993): the running statistics
    are momentum * old + (1 - momentum) * new on EVERY call after the first, and transform() divides by the running mag."""
    from genrl_amd.agent.dreamer_utils import StreamNorm
    g = torch.Generator().manual_seed(7)
    sn = StreamNorm(momentum=momentum, scale=1.0, eps=1e-8, device='cuda')
    mag = mean = sq = None
    for i in range(4):
        x = torch.randn(17, 256, 1, generator=g) * (1.0 + i) + 0.3 * i
        # the reference's recurrence, restated on the CPU
        nm, nmean, nsq = x.abs().mean(), x.mean(), (x * x).mean()
        mag = nm if mag is None else momentum * mag + (1 - momentum) * nm
        mean = nmean if mean is None else momentum * mean + (1 - momentum) * nmean
        sq = nsq if sq is None else momentum * sq + (1 - momentum) * nsq
        ref_out = x if momentum == 1 else x / (mag + 1e-8)
        out, mets = sn(x.cuda())
        assert torch.allclose(sn.mag.cpu(), mag, rtol=3e-6), (i, sn.mag, mag)
        assert torch.allclose(sn.mean.cpu(), mean, rtol=1e-5, atol=1e-6) and torch.allclose(sn.square_mean.cpu(), sq, rtol=3e-6)
        assert torch.allclose(out.cpu(), ref_out, rtol=3e-6, atol=1e-7)
        assert torch.allclose(mets['mean'].cpu(), x.mean(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(mets['normed_std'].cpu(), ref_out.std(), rtol=1e-5)
