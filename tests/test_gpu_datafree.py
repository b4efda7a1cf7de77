"""BASELINE configs[4] (data-free RL, train.py:283-340: no frames, latent starts from the uniform
prior mixed with connector rollouts of random embeddings, random-action and policy warm-up
rollouts, then the imagination update with horizon 15) against outputs of the reference itself
(tests/golden/c5_datafree_tiny.npz, tests/golden/make_datafree_golden.py; RNG draws recorded)."""
import os
import numpy as np
import pytest
import torch

import detgen
from param_shapes import agent_param_shapes
from oracle import genrl_oracle as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'c5_datafree_tiny.npz'))
BS, BL, A, S, K, WARM, H, SEED = [int(x) for x in G['meta']]


class FakeClip:
    ignores_text = True

    def get_txt_feat(self, text):
        g = torch.Generator().manual_seed(123)
        return torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)


def test_data_free_iteration_matches_reference():
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    from genrl_amd import config, noise as gnoise
    zero = dict(lr=0.0, wd=0.0)
    cfg = config.default_cfg(BS, BL, device='cuda', imag_horizon=H, model_opt=zero, actor_opt=zero, critic_opt=zero,
                             **config.tiny_overrides())
    ag = config.make_agent(cfg, act_dim=A)
    ocfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H, deter=32, hidden=32, units=32, cnn_depth=4)
    ag.load_state_dict({k: v.cuda() for k, v in detgen.det_state_dict(agent_param_shapes(ocfg), SEED).items()})
    ag.wm.viclip_model = FakeClip()
    keys = sorted(k for k in G.files if k.startswith('tape.'))
    tp = [torch.from_numpy(G[k]) for k in keys]
    kinds = [k.rsplit('.', 1)[1] for k in keys]
    nw = int(G['n_warm'])
    assert nw == 34 and len(tp) == 66 and kinds[23] == 'normal' and kinds[34] == 'normal'
    sites = {'rssm.unif': [tp[0]], 'imag.target_init_q': [tp[1], tp[65]], 'rssm.prior': tp[2:23],
             'imag.act_eps': [torch.stack(tp[24:34:2]), torch.stack(tp[35:65:2])],       # (the reference's throw-away
             'imag.step_q': [torch.stack(tp[25:34:2]), torch.stack(tp[36:65:2])]}        #  policy samples 23, 34 are dropped)
    dev = 'cuda'
    video_embed = torch.from_numpy(G['video_embed']).to(dev)
    mixmask = torch.from_numpy(G['mixmask']).to(dev)
    fake_action = torch.from_numpy(G['fake_action']).to(dev)
    wm = ag.wm
    n_half = BS * (BL // 2)
    T = wm.connector.n_frames * 2
    B = n_half // T
    with gnoise.inject(sites):
        with torch.no_grad():
            init = wm.rssm.initial(n_half)
            unif = wm.rssm.get_unif_dist(init)
            init['logit'] = unif.mean
            init['stoch'] = unif.sample()
            vinit = wm.connector.video_imagine(video_embed, dreamer_init=None, sample=True, reset_every_n_frames=False,
                                               denoise=True)
            vinit = {k: v.reshape(B * T, *v.shape[2:]) for k, v in vinit.items()}
            init['stoch'] = (mixmask * init['stoch']) + ((~mixmask) * vinit['stoch'])
            post1 = wm.rssm.imagine(fake_action, init, sample=True)
            post1 = {k: v[:, -1].reshape([BS, BL // 2] + list(v.shape[2:])) for k, v in post1.items()}
            init2 = {k: v.reshape([BS, BL // 2] + list(v.shape[1:])) for k, v in init.items()}
            post2 = wm.imagine(ag._imag_behavior.actor, init2, None, WARM)
            post2 = {k: v[-1, :].reshape([BS, BL // 2] + list(v.shape[2:])) for k, v in post2.items() if k in post1}
            post = {k: torch.cat([post1[k], post2[k]], dim=1) for k in post1}
        assert (post['stoch'].argmax(-1).cpu().numpy() == G['post.stoch_idx']).all()
        np.testing.assert_allclose(post['deter'].cpu().numpy(), G['post.deter'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(post['logit'].cpu().numpy(), G['post.logit'], rtol=1e-4, atol=1e-5)
        outputs = dict(post=post, is_terminal=torch.zeros(BS, BL, device=dev))
        _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics={}, seq_data=None)
    n = 0
    for k in G.files:
        if k.startswith('metrics.'):
            name = k[len('metrics.'):]
            np.testing.assert_allclose(float(mets[name]), float(G[k]), rtol=2e-4, atol=2e-6, err_msg=name)
            n += 1
    assert n >= 15, n
