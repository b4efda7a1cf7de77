"""EnsembleRSSM.observe with the posterior inside the recurrence (`single_obs_posterior: false`, conf/defaults/dreamer_v3.yaml:5;
agent/dreamer_utils.py:362-371, 432-457) as ONE scan node (ops.observe_seq, csrc/seq.hip: genrl_observe_seq_fwd / _bwd) against
the step-by-step form it replaces (EnsembleRSSM._observe_stepwise: one autograd node per layer and step, itself pinned to the
reference's c3 golden and to the oracle by tests/test_gpu_iteration.py / test_gpu_fullsize.py): posterior / prior states and
every gradient -- parameters, embedding, initial state -- on the same noise, with is_first resets inside the window; the C launch
loops against their Python twins bit for bit; run-to-run determinism."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu

# (T, B, S, K, deter, hidden, embed, A)
CASES = [(6, 2, 4, 4, 32, 32, 24, 6), (9, 3, 8, 8, 64, 48, 40, 10), (5, 8, 32, 32, 512, 512, 1536, 6), (4, 40, 32, 32, 512, 512, 1536, 6),
         (3, 4, 32, 32, 1024, 1024, 1536, 10)]


def _rssm(S, K, D, U, E, A, seed):
    from genrl_amd.agent import dreamer_utils as common
    torch.manual_seed(seed)
    r = common.EnsembleRSSM(ensemble=1, stoch=S, deter=D, hidden=U, discrete=K, act='SiLU', norm='layer', action_dim=A, embed_dim=E,
                            device='cuda', single_obs_posterior=False).cuda()
    with torch.no_grad():
        for n, p in r.named_parameters():      # LayerNorm parameters away from (1, 0), biases away from 0
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    return r


def _run(case, mode, seed=0, with_state=True, fuse=True):
    from genrl_amd import noise as gnoise
    T, B, S, K, D, U, E, A = case
    r = _rssm(S, K, D, U, E, A, seed)
    g = torch.Generator(device='cuda').manual_seed(seed + 1)
    rn = lambda *s: torch.randn(*s, device='cuda', generator=g)
    embed = rn(B, T, E).requires_grad_(True)
    action = torch.tanh(rn(B, T, A))
    is_first = torch.zeros(B, T, dtype=torch.bool, device='cuda')
    is_first[:, 0] = True
    if T > 2:
        is_first[B // 2, T // 2] = True
    state = None
    if with_state:
        st = torch.nn.functional.one_hot(torch.randint(0, K, (B, S), device='cuda', generator=g), K).float().requires_grad_(True)
        state = {'stoch': st, 'deter': (0.5 * rn(B, D)).requires_grad_(True), 'logit': rn(B, S, K)}
        is_first[:, 0] = False
        is_first[0, 0] = True
    qs = [torch.empty(B * S, K, device='cuda').exponential_(1.0, generator=g) for _ in range(2 * T)]
    sites = {'rssm.post': qs[:T], 'rssm.prior': qs[T:]}
    w = {k: rn(B, T, *sh) for k, sh in (('ps', (S, K)), ('pl', (S, K)), ('d', (D,)), ('qs', (S, K)), ('ql', (S, K)))}
    env = {'GENRL_OBSERVE_SEQ': '0' if mode == 'stepwise' else '1', 'GENRL_SEQ_C': '0' if mode == 'python' else '1'}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    from genrl_amd import ops
    seq_c, fuse_was = ops.SEQ_C, ops.OBSERVE_FUSE
    ops.SEQ_C = mode != 'python'
    ops.OBSERVE_FUSE = fuse
    try:
        with gnoise.inject(sites):
            post, prior = r.observe(embed, action, is_first, state)
        loss = ((post['stoch'] * w['ps']).sum() + (post['logit'] * w['pl']).sum() + (post['deter'] * w['d']).sum()
                + (prior['stoch'] * w['qs']).sum() + (prior['logit'] * w['ql']).sum())
        loss.backward()
    finally:
        ops.SEQ_C, ops.OBSERVE_FUSE = seq_c, fuse_was
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    torch.cuda.synchronize()
    out = {f'post.{k}': v.detach() for k, v in post.items()}
    out.update({f'prior.{k}': v.detach() for k, v in prior.items()})
    out['d.embed'] = embed.grad.detach()
    if with_state:
        out['d.stoch0'] = state['stoch'].grad.detach()
        out['d.deter0'] = state['deter'].grad.detach()
    out.update({f'd.{n}': p.grad.detach() for n, p in r.named_parameters() if p.grad is not None})
    return out


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('with_state', [False, True])
@pytest.mark.parametrize('fuse', [False, True])
def test_observe_scan_matches_the_stepwise_form(case, with_state, fuse):
    """fuse: the six-launch forward (the one-hot latent's product as a gather fused with its LayerNorm, the head product with the sample
    as its epilogue: genrl_onehot_gather_ln_fwd, genrl_linear_sample32) against the same stepwise reference"""
    ref = _run(case, 'stepwise', seed=sum(case), with_state=with_state)
    got = _run(case, 'scan', seed=sum(case), with_state=with_state, fuse=fuse)
    assert set(ref) == set(got), set(ref) ^ set(got)
    for k in ('post.stoch', 'prior.stoch'):
        assert torch.equal(got[k], ref[k]), k                    # sampled latents: exact
    for k, b in ref.items():
        a = got[k]
        scale = b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item() / scale
        assert torch.isfinite(a).all() and err <= (2e-5 if not k.startswith('d.') else 2e-4), (k, err)


@pytest.mark.parametrize('case', [CASES[0], CASES[2]])
def test_observe_scan_c_loop_is_the_python_loop(case):
    a = _run(case, 'scan', seed=3)
    b = _run(case, 'python', seed=3)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_observe_scan_is_bit_reproducible():
    a = _run(CASES[3], 'scan', seed=11)
    b = _run(CASES[3], 'scan', seed=11)
    for k in a:
        assert torch.equal(a[k], b[k]), k
