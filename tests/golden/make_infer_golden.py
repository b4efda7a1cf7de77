"""Golden vectors for the inference rows of SURVEY §8f: `GenRLAgent.report` (decoded frames of
`video_pred` and of the connector's text/video-conditioned prediction, agent/genrl.py:64-106,
agent/dreamer.py:99-109,307-321) and `DreamerAgent.act` (agent/dreamer.py:41-64), produced by the
reference itself at tiny dims.  The reference's RNG draws are RECORDED (ref_harness.NoiseTape) and
stored, so the HIP path can consume exactly the same noise.  Authoring container only."""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import detgen                                                      # noqa: E402
import ref_harness as rh                                           # noqa: E402

B, T, A, SEED, NVID = 4, 24, 10, 5, 3


def main():
    assert rh.available()
    ag = rh.make_ref_agent(B, T, A=A, imag_reward_fn='video_text_reward', **detgen.tiny_overrides())
    ag.wm.viclip_model = rh.FakeClip()
    sd = ag.state_dict()
    for bh in (ag._imag_behavior, ag._acting_behavior):
        for d_ in bh._target_critic.parameters():
            d_.data = d_.data.clone()
    ag.load_state_dict(detgen.det_state_dict({k: v.shape for k, v in sd.items()}, SEED))
    batch = detgen.det_batch(B, T, A=A, img=64, seed=SEED)
    tb = rh.to_torch(batch)
    out = {'meta': np.array([B, T, A, ag.cfg.rssm.stoch, ag.cfg.rssm.discrete, NVID, SEED])}
    # ---- report
    tape = rh.NoiseTape('record')
    with rh.inject_noise(tape), torch.no_grad():
        rep = ag.report(tb, nvid=NVID) if 'nvid' in ag.report.__code__.co_varnames else ag.report(tb)
    out['n_labels'] = np.array(rep['text_to_video'].shape[0])
    rep['text_to_video'] = rep['text_to_video'][:2]          # every prompt maps to the same stub embedding
    for k, v in rep.items():                                  # videos in [0,1]: every 8th pixel + per-frame channel means
        v = v.numpy().astype(np.float32)
        out[f'report.{k}.sub'] = v[..., ::8, ::8].copy()
        out[f'report.{k}.mean'] = v.astype(np.float64).mean((-1, -2)).astype(np.float32)
    for i, (kind, x) in enumerate(tape.tape):
        out[f'report_tape.{i:03d}.{kind}'] = x.numpy()
    # ---- act: 4 consecutive steps, carrying the state; eval (mean action, mode-free latent sampling per
    # cfg.eval_state_mean) and training (sampled) flavours
    for flavour, eval_mode in (('eval', True), ('train', False)):
        tape = rh.NoiseTape('record')
        state = None
        with rh.inject_noise(tape):
            for t in range(4):
                obs = {k: v[0, t] for k, v in batch.items() if k != 'action'}
                action, state = ag.act(obs, None, t, eval_mode, state)
                out[f'act_{flavour}.action{t}'] = np.asarray(action, np.float32)
                out[f'act_{flavour}.stoch_idx{t}'] = state[0]['stoch'].argmax(-1).numpy().astype(np.int16)
                out[f'act_{flavour}.deter{t}'] = state[0]['deter'].detach().numpy()
        for i, (kind, x) in enumerate(tape.tape):
            out[f'act_{flavour}_tape.{i:03d}.{kind}'] = x.numpy()
    np.savez_compressed(os.path.join(HERE, 'infer_tiny.npz'), **out)
    print('wrote infer_tiny.npz', {k: v.shape for k, v in out.items() if k.startswith('report.')},
          'tape', len([k for k in out if k.startswith('report_tape')]), sum(v.nbytes for v in out.values()) // 1024, 'KB')


if __name__ == '__main__':
    main()
