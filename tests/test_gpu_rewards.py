"""Every branch of the reward functions (tools/genrl_utils.py:250-409) against the reference's own outputs
(tests/golden/rewards.npz, written by tests/golden/make_reward_golden.py from the imported reference): the seven score functions
of compute_reward x {no alignment, align_initial, align_sequence} x weighted_align, and video_video_reward behind a stub video
embedding.  Rewards and the gradients of a fixed weighted sum w.r.t. the agent's stoch / logit."""
import os
import numpy as np
import pytest
import torch

import detgen
from oracle import genrl_oracle as O
from param_shapes import agent_param_shapes

pytestmark = pytest.mark.gpu
G = dict(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'rewards.npz')))
SCORES = ('cosine', 'max_cosine', 'neg_mse', 'exp_neg_mse', 'neg_kl', 'max_like', 'combo')
CASES = [(s, m, w) for s in SCORES for m in ('none', 'initial', 'sequence') for w in (0, 1) if not (w and m == 'none')]


class FakeClip:
    ignores_text = True

    def get_txt_feat(self, text):
        g = torch.Generator().manual_seed(123)
        return torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)


@pytest.fixture(scope='module')
def agent():
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    from genrl_amd import config
    T, B, S, K, D, seed = [int(x) for x in G['meta']]
    cfg = config.default_cfg(2, 16, device='cuda', **config.tiny_overrides())
    ag = config.make_agent(cfg, act_dim=10)
    ocfg = O.make_cfg(stoch=S, discrete=K, deter=D, hidden=D, units=32, cnn_depth=4, act_dim=10)
    p = detgen.det_state_dict(agent_param_shapes(ocfg), seed)
    ag.load_state_dict({k: v.cuda() for k, v in p.items()})
    ag.wm.viclip_model = FakeClip()
    return ag


def _seq():
    return {k: torch.from_numpy(G[f'seq.{k}']).cuda().requires_grad_(k != 'deter') for k in ('stoch', 'logit', 'deter')}


@pytest.mark.parametrize('score,mode,weighted', CASES)
def test_reward_branch_matches_reference(agent, score, mode, weighted):
    from genrl_amd.tools import genrl_utils as gu
    key = f'{score}.{mode}.{weighted}'
    assert f'{key}.reward' in G, 'the reference ran every branch when the fixture was made'
    agent.unconditional_target = {k: torch.from_numpy(G[f'target.{k}']).cuda() for k in ('stoch', 'logit', 'deter')}
    for stale in ('_target_stoch_planes',):
        if hasattr(agent, stale):
            delattr(agent, stale)
    seq = _seq()
    try:
        r = gu.video_text_reward(agent, seq, score_fn=score, weighted_align=bool(weighted), align_initial=mode == 'initial',
                                 align_sequence=mode == 'sequence')
        w = torch.from_numpy(G['weight']).cuda()
        (r * w).sum().backward()
    finally:
        del agent.unconditional_target
        if hasattr(agent, '_target_stoch_planes'):
            del agent._target_stoch_planes
    ref = G[f'{key}.reward']
    np.testing.assert_allclose(r.detach().cpu().numpy(), ref, rtol=3e-5, atol=3e-6, err_msg=key)
    for name in ('stoch', 'logit'):
        gref = G[f'{key}.d{name}']
        got = seq[name].grad
        if gref.size == 0:
            assert got is None or float(got.abs().max()) == 0.0, (key, name)
            continue
        assert got is not None, (key, name)
        scale = np.abs(gref).max() + 1e-12
        assert np.abs(got.cpu().numpy() - gref).max() <= 1e-4 * scale + 1e-7, (key, name, np.abs(got.cpu().numpy() - gref).max(), scale)


def test_video_video_reward_matches_reference(agent):
    """ref :372-409 behind the (host-side, stubbed) video embedding: connector.video_imagine -> cached target ->
    video_text_reward(max_cosine, align_sequence, skip_first_target)."""
    from genrl_amd import noise
    from genrl_amd.tools import genrl_utils as gu
    for stale in ('unconditional_target', '_target_stoch_planes'):
        if hasattr(agent, stale):
            delattr(agent, stale)
    agent.wm.video_prompt_embed = torch.from_numpy(G['vv.embed'])
    seq = _seq()
    try:
        with noise.inject({'imag.target_init_q': torch.from_numpy(G['vv.init_q'])}):
            r = gu.video_video_reward(agent, seq, score_fn='max_cosine', sample_for_target=False, skip_first_target=True,
                                      align_sequence=True)
        (r * torch.from_numpy(G['weight']).cuda()).sum().backward()
        tidx = agent.unconditional_target['stoch'].argmax(-1).cpu().numpy()
    finally:
        for stale in ('unconditional_target', '_target_stoch_planes'):
            if hasattr(agent, stale):
                delattr(agent, stale)
        del agent.wm.video_prompt_embed
    assert np.array_equal(tidx, G['vv.target_idx'])
    np.testing.assert_allclose(r.detach().cpu().numpy(), G['vv.reward'], rtol=3e-5, atol=3e-6)
    gref = G['vv.dstoch']
    assert np.abs(seq['stoch'].grad.cpu().numpy() - gref).max() <= 1e-4 * (np.abs(gref).max() + 1e-12) + 1e-7
