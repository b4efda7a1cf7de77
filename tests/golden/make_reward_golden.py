"""Golden vectors for EVERY branch of the reference's reward functions (tools/genrl_utils.py:250-409): compute_reward's seven
score functions x {no alignment, align_initial, align_sequence} x {weighted_align off, on}, and video_video_reward with a stub
video embedding.  Run in the authoring container only:

    python tests/golden/make_reward_golden.py

The reference is imported through ref_harness.py (nothing of it is copied): a tiny-dims GenRLAgent with deterministic per-name
weights (tests/detgen.py), a synthetic imagined sequence and a synthetic cached target; stored are the inputs, the rewards and
the gradients of a fixed weighted sum of the rewards w.r.t. the agent's stoch and logit."""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh
import detgen

SCORES = ('cosine', 'max_cosine', 'neg_mse', 'exp_neg_mse', 'neg_kl', 'max_like', 'combo')
MODES = ('none', 'initial', 'sequence')
T, B, SEED = 12, 6, 5


def inputs(S, K, D):
    g = torch.Generator().manual_seed(SEED)
    onehot = lambda: torch.nn.functional.one_hot(torch.randint(0, K, (T, B, S), generator=g), K).float()
    seq = dict(stoch=onehot(), logit=torch.randn(T, B, S, K, generator=g) * 1.5, deter=torch.randn(T, B, D, generator=g))
    tgt = dict(stoch=onehot(), logit=torch.randn(T, B, S, K, generator=g) * 1.5, deter=torch.randn(T, B, D, generator=g))
    # make a few agent latents coincide with the target's (exact ties in max-cosine norms)
    seq['stoch'][3:5] = tgt['stoch'][0:2]
    w = torch.rand(T, B, 1, generator=g) + 0.5
    return seq, tgt, w


def main():
    m = rh.ref_modules()
    ag = rh.make_ref_agent(2, 16, A=10, imag_reward_fn="video_text_reward", **detgen.tiny_overrides())
    for bh in (ag._imag_behavior, ag._acting_behavior):
        for d_ in bh._target_critic.parameters():
            d_.data = d_.data.clone()
    ag.load_state_dict(detgen.det_state_dict({k: v.shape for k, v in ag.state_dict().items()}, SEED))
    S, K, D = ag.cfg.rssm.stoch, ag.cfg.rssm.discrete, ag.cfg.rssm.deter
    seq0, tgt, w = inputs(S, K, D)
    out = {f'seq.{k}': v.numpy() for k, v in seq0.items()}
    out.update({f'target.{k}': v.numpy() for k, v in tgt.items()})
    out['weight'] = w.numpy()
    n = 0
    for score in SCORES:
        for mode in MODES:
            for weighted in (False, True):
                if weighted and mode == 'none':
                    continue
                seq = {k: v.clone().requires_grad_(k != 'deter') for k, v in seq0.items()}
                ag.unconditional_target = {k: v.clone() for k, v in tgt.items()}
                key = f'{score}.{mode}.{int(weighted)}'
                try:
                    r = m.gu.video_text_reward(ag, seq, score_fn=score, weighted_align=weighted, align_initial=mode == 'initial',
                                               align_sequence=mode == 'sequence')
                except Exception as e:      # a branch the reference itself cannot run: recorded as such
                    out[f'{key}.error'] = np.array(type(e).__name__)
                    print(key, 'reference raises', type(e).__name__, e)
                    continue
                assert r.shape == (T, B, 1), r.shape
                (r * w).sum().backward()
                out[f'{key}.reward'] = r.detach().numpy()
                for name in ('stoch', 'logit'):
                    gten = seq[name].grad
                    out[f'{key}.d{name}'] = gten.numpy() if gten is not None else np.zeros(0, np.float32)
                n += 1
    # video_video_reward: the host-side video decoding / embedding is stubbed (cv2 + InternVideo2 are absent); what is pinned is
    # the path behind the embedding -- connector.video_imagine -> cached target -> video_text_reward
    del ag.unconditional_target
    g = torch.Generator().manual_seed(77)
    vfeat = torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)
    gu = m.gu
    gu.TASK2VIDEO = dict(getattr(gu, 'TASK2VIDEO', {}), stickman_walk='stub.mp4')
    gu.cv2.VideoCapture = lambda path: None
    gu.cv2.cvtColor = lambda x, code: x
    gu.cv2.COLOR_BGR2RGB = 0
    gu._frame_from_video = lambda video: iter([np.zeros((4, 4, 3), np.uint8)] * 2)
    gu.get_video_feat = lambda frames, clip, flip=False: (vfeat.clone(), None)
    ag.wm.viclip_model = rh.FakeClip()
    tape = rh.NoiseTape('record')
    seq = {k: v.clone().requires_grad_(k != 'deter') for k, v in seq0.items()}
    with rh.inject_noise(tape):
        r = gu.video_video_reward(ag, seq, score_fn='max_cosine', sample_for_target=False, skip_first_target=True,
                                  align_sequence=True)
    (r * w).sum().backward()
    assert len(tape.tape) == 1 and tape.tape[0][0] == 'exp', [(k, tuple(x.shape)) for k, x in tape.tape]
    out['vv.embed'] = vfeat.numpy(); out['vv.init_q'] = tape.tape[0][1].numpy()
    out['vv.reward'] = r.detach().numpy(); out['vv.dstoch'] = seq['stoch'].grad.numpy()
    out['vv.target_idx'] = ag.unconditional_target['stoch'].argmax(-1).to(torch.int16).numpy()
    out['meta'] = np.array([T, B, S, K, D, SEED]); out['torch_version'] = np.array(torch.__version__)
    np.savez_compressed(f'{HERE}/rewards.npz', **out)
    print('rewards.npz', n, 'branches +', 'video_video_reward')


if __name__ == '__main__':
    main()
