"""The state-resident GRU scan (csrc/scan_coop.hip: one persistent launch per sequence, W_h resident in LDS, grid barriers
between the steps) against the per-step launches it replaces (ops._GRUSeq's default path, itself pinned to the reference's
goldens): states, LayerNorm statistics and -- through the unchanged backward, which reads what the forward leaves -- every
gradient.  Both variants (two barriers per step / one barrier with redundant gates), with and without the is_first mask
(agent/dreamer_utils.py:432-440), tiny and full width."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [(16, 4, 32, 32, 1), (16, 4, 32, 32, 2), (7, 8, 48, 64, 1), (9, 16, 40, 64, 2), (32, 32, 1024, 1024, 2),
         (32, 4, 1024, 1024, 1), (32, 4, 1024, 1024, 2), (12, 8, 520, 1024, 1)]


def _run(T, B, I, D, masked, variant, seed):
    from genrl_amd import ops
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = (torch.randn(T, B, I, device='cuda', generator=g)).requires_grad_(True)
    h0 = (torch.randn(B, D, device='cuda', generator=g) * 0.5).requires_grad_(True)
    W = (torch.randn(3 * D, I + D, device='cuda', generator=g) / (I + D) ** 0.5).requires_grad_(True)
    gamma = (1.0 + 0.1 * torch.randn(3 * D, device='cuda', generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(3 * D, device='cuda', generator=g)).requires_grad_(True)
    mask = None
    if masked:
        mask = torch.ones(T, B, device='cuda')
        mask[0] = 0.0; mask[T // 2, B // 2] = 0.0
    wout = torch.randn(T, B, D, device='cuda', generator=g)
    os.environ['GENRL_SCAN_COOP'] = str(variant)
    try:
        out = ops.gru_seq(x, mask, h0, W, gamma, beta)
        (out * wout).sum().backward()
    finally:
        os.environ.pop('GENRL_SCAN_COOP', None)
    torch.cuda.synchronize()
    return [out.detach()] + [t.grad.detach() for t in (x, h0, W, gamma, beta)]


@pytest.mark.parametrize('T,B,I,D,variant', CASES)
@pytest.mark.parametrize('masked', [False, True])
def test_persistent_scan_matches_per_step_launches(T, B, I, D, variant, masked):
    ref = _run(T, B, I, D, masked, 0, seed=T + B + D)
    got = _run(T, B, I, D, masked, variant, seed=T + B + D)
    names = ('out', 'dx', 'dh0', 'dW', 'dgamma', 'dbeta')
    for n, a, b in zip(names, got, ref):
        scale = b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item() / scale
        assert torch.isfinite(a).all() and err <= (2e-5 if n == 'out' else 2e-4), (n, err)


def test_persistent_scan_is_bit_reproducible():
    a = _run(32, 4, 1024, 1024, True, 1, seed=5)
    b = _run(32, 4, 1024, 1024, True, 1, seed=5)
    c = _run(32, 32, 1024, 1024, True, 2, seed=6)
    d = _run(32, 32, 1024, 1024, True, 2, seed=6)
    for u, v in list(zip(a, b)) + list(zip(c, d)):
        assert torch.equal(u, v)
