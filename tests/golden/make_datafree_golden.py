"""Golden vectors for BASELINE configs[4] (data-free RL, train.py:283-340 with train_from_data=False,
start_from_video='mix', mix_random_actions=True): the block's call sequence is replayed on the
REFERENCE agent (rssm.initial / get_unif_dist / connector.video_imagine / rssm.imagine /
wm.imagine warm-up / update_imag_behavior with imag_horizon 15) at tiny dims with its RNG draws
recorded; the random tensors train.py draws with torch.randn/rand directly are generated here and
stored.  Authoring container only."""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import detgen                                                      # noqa: E402
import ref_harness as rh                                           # noqa: E402

BS, BL, A, SEED, WARM, H = 4, 16, 10, 11, 5, 15


def main():
    assert rh.available()
    zero = dict(lr=0.0, wd=0.0)
    ag = rh.make_ref_agent(BS, BL, A=A, imag_reward_fn='video_text_reward', imag_horizon=H, model_opt=zero,
                           actor_opt=zero, critic_opt=zero, **detgen.tiny_overrides())
    ag.wm.viclip_model = rh.FakeClip()
    sd = ag.state_dict()
    for bh in (ag._imag_behavior, ag._acting_behavior):
        for d_ in bh._target_critic.parameters():
            d_.data = d_.data.clone()
    ag.load_state_dict(detgen.det_state_dict({k: v.shape for k, v in sd.items()}, SEED))
    g = torch.Generator().manual_seed(SEED)
    n_half = BS * (BL // 2)
    T = ag.wm.connector.n_frames * 2
    B = n_half // T
    E = ag.wm.connector.viclip_emb_dim
    video_embed = torch.nn.functional.normalize(torch.randn(B, T, E, generator=g), dim=-1)
    mixmask = torch.rand(B * T, 1, 1, generator=g) > 0.5
    fake_action = torch.rand(n_half, WARM, A, generator=g) * 2 - 1
    out = {'meta': np.array([BS, BL, A, ag.cfg.rssm.stoch, ag.cfg.rssm.discrete, WARM, H, SEED]),
           'video_embed': video_embed.numpy(), 'mixmask': mixmask.numpy(), 'fake_action': fake_action.numpy()}
    tape = rh.NoiseTape('record')
    wm = ag.wm
    with rh.inject_noise(tape):
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
            post2 = {k: v[-1, :].reshape([BS, BL // 2] + list(v.shape[2:])) for k, v in post2.items()}
            post = {k: torch.cat([post1[k], post2[k]], dim=1) for k in post1}
        n_warm = len(tape.tape)
        outputs = dict(post=post, is_terminal=torch.zeros(BS, BL))
        _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics={}, seq_data=None)
    out['n_warm'] = np.array(n_warm)
    out['post.stoch_idx'] = post['stoch'].argmax(-1).numpy().astype(np.int16)
    out['post.deter'] = post['deter'].numpy()
    out['post.logit'] = post['logit'].numpy()
    for k, v in mets.items():
        out[f'metrics.{k}'] = np.asarray(float(v), np.float64)
    for i, (kind, x) in enumerate(tape.tape):
        out[f'tape.{i:03d}.{kind}'] = x.numpy()
    np.savez_compressed(os.path.join(HERE, 'c5_datafree_tiny.npz'), **out)
    print('wrote c5_datafree_tiny.npz; tape', len(tape.tape), 'warm', n_warm, 'KB', sum(v.nbytes for v in out.values()) // 1024)
    print([(k, x.shape) for k, x in tape.tape][:6], '...', [(k, tuple(x.shape)) for k, x in tape.tape][n_warm - 12:n_warm + 3])


if __name__ == '__main__':
    main()
