"""Inference rows of SURVEY §8f against the reference's own outputs (tests/golden/infer_tiny.npz,
made by tests/golden/make_infer_golden.py with the reference's RNG draws recorded):
 * GenRLAgent.report — decoded frames of the open-loop world-model prediction (`video_pred`), of the
   text-to-video rollouts (`report_text2video`) and of the connector's video-conditioned prediction;
 * DreamerAgent.act — filtered latent + action over consecutive environment steps (eval and
   exploration modes).
The HIP path consumes the recorded noise through its named sites; sampled latent indices must be
identical, frames (values in [0,1]) within 1e-4 absolute."""
import os
import numpy as np
import pytest
import torch

import detgen
from param_shapes import agent_param_shapes
from oracle import genrl_oracle as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'infer_tiny.npz'))
B, T, A, S, K, NVID, SEED = [int(x) for x in G['meta']]


class FakeClip:
    ignores_text = True

    def get_txt_feat(self, text):
        g = torch.Generator().manual_seed(123)
        return torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)


def tape(prefix):
    keys = sorted(k for k in G.files if k.startswith(prefix + '.'))
    return [(k.rsplit('.', 1)[1], torch.from_numpy(G[k])) for k in keys]


@pytest.fixture(scope='module')
def agent():
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    from genrl_amd import config
    from genrl_amd.tools import genrl_utils as GU
    cfg = config.default_cfg(B, T, device='cuda', **config.tiny_overrides())
    cfg['additional_report_fns'] = ['report_text2video']
    ag = config.make_agent(cfg, act_dim=A)
    ocfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, deter=32, hidden=32, units=32, cnn_depth=4)
    ag.load_state_dict({k: v.cuda() for k, v in detgen.det_state_dict(agent_param_shapes(ocfg), SEED).items()})
    ag.wm.viclip_model = FakeClip()
    GU.DOMAIN2PREDICATES['stickman'] = [f'behaviour {i}' for i in range(int(G['n_labels']))]
    return ag


def test_report_frames_match_reference(agent):
    from genrl_amd import noise as gnoise
    tp = tape('report_tape')
    rows1, rows2, L = B * S, NVID * S, int(G['n_labels'])
    assert len(tp) == 10 + (T - 5) + 1 + 16 + 1 + (T - 8)
    ex = [x for _, x in tp]
    sites = {'wm.prior_q': [torch.stack(ex[0:10:2]), torch.stack(ex[30:46:2])],
             'wm.post_q': [torch.stack(ex[1:10:2]), torch.stack(ex[31:46:2])],
             'rssm.prior': ex[10:10 + T - 5] + ex[47:],
             'imag.target_init_q': [ex[29], ex[46]]}
    assert ex[29].shape == (L * S, K) and ex[46].shape == (rows2, K) and ex[0].shape == (rows1, K)
    batch = {k: torch.from_numpy(v).cuda() for k, v in detgen.det_batch(B, T, A=A, img=64, seed=SEED).items()}
    with gnoise.inject(sites):
        rep = agent.report(batch, nvid=NVID)
    for name in ('openl_observation', 'text_to_video', 'video_clip_pred'):
        got = rep[name].float().cpu().numpy()
        if name == 'text_to_video':
            assert got.shape[0] == L
            got = got[:2]
        assert got.min() >= -1e-6 and got.max() <= 1 + 1e-6 or name == 'text_to_video'
        np.testing.assert_allclose(got[..., ::8, ::8], G[f'report.{name}.sub'], atol=1e-4, rtol=0, err_msg=name)
        np.testing.assert_allclose(got.astype(np.float64).mean((-1, -2)), G[f'report.{name}.mean'], atol=2e-5, rtol=0,
                                   err_msg=name)


@pytest.mark.parametrize('flavour,eval_mode', [('eval', True), ('train', False)])
def test_act_matches_reference(agent, flavour, eval_mode):
    from genrl_amd import noise as gnoise
    tp = tape(f'act_{flavour}_tape')
    exps = [x for k, x in tp if k == 'exp']
    sites = {'rssm.prior': exps[0::2], 'rssm.post': exps[1::2], 'actor': [x for k, x in tp if k == 'normal']}
    batch = detgen.det_batch(B, T, A=A, img=64, seed=SEED)
    state = None
    with gnoise.inject(sites):
        for t in range(4):
            obs = {k: v[0, t] for k, v in batch.items() if k != 'action'}
            action, state = agent.act(obs, None, t, eval_mode, state)
            assert (state[0]['stoch'].argmax(-1).cpu().numpy() == G[f'act_{flavour}.stoch_idx{t}']).all()
            np.testing.assert_allclose(state[0]['deter'].cpu().numpy(), G[f'act_{flavour}.deter{t}'], rtol=1e-4, atol=1e-5)
            np.testing.assert_allclose(action, G[f'act_{flavour}.action{t}'], rtol=1e-4, atol=1e-5)
