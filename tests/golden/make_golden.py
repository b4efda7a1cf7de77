"""Generate golden vectors from the reference (run in the authoring container only):

    python tests/golden/make_golden.py

Imports mazpie/genrl from /root/reference through tests/golden/ref_harness.py, loads
deterministic per-name weights (tests/detgen.py), replays deterministic noise through the
reference's RNG call sites and stores inputs + the reference's outputs/gradients as .npz.
No reference source is copied; fixtures are data only.  Generated with torch (see meta)."""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as rh
import detgen

torch.set_num_threads(8)


def flat(prefix, d, out):
    for k, v in d.items():
        if isinstance(v, dict):
            flat(f'{prefix}{k}.', v, out)
        else:
            out[f'{prefix}{k}'] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


def summarize(t, k=8):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()] + t[:k].tolist())


def kat():
    m = rh.ref_modules(); C = m.common
    g = torch.Generator().manual_seed(7)
    out = {}
    r = torch.rand(6, 5, 1, generator=g) * 2; v = torch.randn(6, 5, 1, generator=g)
    d = 0.99 * torch.ones(6, 5, 1); boot = torch.randn(5, 1, generator=g)
    out.update(lr_reward=r, lr_value=v, lr_disc=d, lr_boot=boot,
               lr_out=C.lambda_return(r, v, d, bootstrap=boot, lambda_=0.95, axis=0))
    logits = torch.randn(7, 3, 255, generator=g)
    x = torch.tensor([-30., -20., -3.3, 0., 1e-3, 0.5, 19.999]).reshape(7, 1, 1).repeat(1, 3, 1) * torch.tensor([1., 0.5, 1.7]).reshape(1, 3, 1)
    th = C.TwoHotDist(logits)
    out.update(th_logits=logits, th_x=x, th_logprob=th.log_prob(x), th_mean=th.mean)
    y = torch.tensor([-5., -1., 0., 0.3, 4.]); out.update(sl_x=y, sl_symlog=C.symlog(y), sl_symexp=C.symexp(y))
    lp, lq = torch.randn(3, 4, 6, 5, generator=g) * 2, torch.randn(3, 4, 6, 5, generator=g) * 2
    import torch.distributions as D
    dp = D.Independent(C.OneHotDist(lp), 1); dq = D.Independent(C.OneHotDist(lq), 1)
    out.update(oh_lp=lp, oh_lq=lq, oh_kl=D.kl_divergence(dp, dq), oh_ent=dp.entropy(),
               oh_probs=C.OneHotDist(lp).probs, oh_mode=C.OneHotDist(lp).mode())
    cell = C.GRUCell(8, 12, norm=True, device='cpu')
    sd = detgen.det_state_dict({k: v.shape for k, v in cell.state_dict().items()}, 3)
    cell.load_state_dict(sd)
    xi, h = torch.randn(5, 8, generator=g), torch.randn(5, 12, generator=g)
    out.update(gru_x=xi, gru_h=h, gru_out=cell(xi, [h])[0], **{f'gru_p.{k}': v for k, v in sd.items()})
    ln = C.ImgChLayerNorm(6); sdl = detgen.det_state_dict({k: v.shape for k, v in ln.state_dict().items()}, 4)
    ln.load_state_dict(sdl); xim = torch.randn(2, 6, 3, 5, generator=g)
    out.update(chln_x=xim, chln_out=ln(xim), **{f'chln_p.{k}': v for k, v in sdl.items()})
    u, w = torch.randn(4, 3, 16, generator=g), torch.randn(4, 3, 16, generator=g) * 2
    out.update(mc_u=u, mc_v=w, mc_out=m.gu.max_cosine_similarity(u, w))
    np.savez_compressed(f'{HERE}/kat.npz', **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print('kat.npz', len(out))


def run_iteration(B, T, A, over, seed, lr_zero, full_tensors, batch, img=64, store_batch=True):
    """One train.py iteration (train.py:273-340) on the reference with injected noise."""
    m = rh.ref_modules()
    over = dict(over)
    if lr_zero:
        for k in ('model_opt', 'actor_opt', 'critic_opt'):
            over[k] = dict(lr=0.0, wd=0.0)
    ag = rh.make_ref_agent(B, T, A=A, img=img, imag_reward_fn='capture_reward', **over)
    ag.wm.viclip_model = rh.FakeClip()
    sd = ag.state_dict()
    det = detgen.det_state_dict({k: v.shape for k, v in sd.items()}, seed)
    for bh in (ag._imag_behavior, ag._acting_behavior):     # un-alias slow critic (agent/dreamer.py:361-362)
        for d_ in bh._target_critic.parameters():
            d_.data = d_.data.clone()
    ag.load_state_dict(det)
    # slow critic starts as a copy of the critic in the reference (agent/dreamer.py:361-362); keep
    # the deterministic independent weights instead so its path is exercised, but stop the hard
    # copy from aliasing: parameters are separate tensors after load_state_dict.
    S, K = ag.cfg.rssm.stoch, ag.cfg.rssm.discrete
    H = ag.cfg.imag_horizon
    noise = detgen.iteration_noise(B, T, S, K, A, H, seed=seed)
    tape = rh.NoiseTape('replay', detgen.tape_from_noise(noise, T, H))
    names = {id(p): n for n, p in ag.named_parameters()}
    grads, cap = {}, {}
    orig_clip = torch.nn.utils.clip_grad_norm_
    phase = ['wm']

    def clip_capture(params, clip, *a, **k):
        params = list(params)
        grads[phase[0]] = {names[id(p)]: p.grad.detach().clone() for p in params if p.grad is not None}
        return orig_clip(params, clip, *a, **k)

    def capture_reward(agent, seq, **kw):
        r = m.gu.video_text_reward(agent, seq, **kw)
        cap['seq'] = {k: v.detach().clone() for k, v in seq.items()}
        cap['reward'] = r.detach().clone()
        cap['target_stoch'] = agent.unconditional_target['stoch'].detach().clone()
        return r
    m.genrl.capture_reward = capture_reward
    orig_target = ag._imag_behavior.target

    def target_capture(seq):
        t, mets, base = orig_target(seq)
        cap['lambda_target'] = t.detach().clone()
        return t, mets, base
    ag._imag_behavior.target = target_capture
    tb = rh.to_torch(batch)
    torch.nn.utils.clip_grad_norm_ = clip_capture
    out = {}
    try:
        with rh.inject_noise(tape):
            # WorldModel.update runs the first connector update inside; split phases via the hook
            orig_detached = ag.wm.update_additional_detached_modules

            def detached_hook(*a, **k):
                phase[0] = 'conn1' if 'conn1' not in grads else 'conn2'
                return orig_detached(*a, **k)
            ag.wm.update_additional_detached_modules = detached_hook
            state, outputs, mets = ag.update_wm(tb, 0)
            mets_wm = {k: torch.as_tensor(v).clone() for k, v in mets.items()}
            _, mets = ag.wm.update_additional_detached_modules(tb, outputs, mets)
            mets_c2 = {k: torch.as_tensor(v).clone() for k, v in mets.items() if 'connector' in k or 'aligner' in k}
            phase[0] = 'actor'
            orig_critic_loss = ag._imag_behavior.critic_loss

            def critic_hook(*a, **k):
                phase[0] = 'critic'
                return orig_critic_loss(*a, **k)
            ag._imag_behavior.critic_loss = critic_hook
            _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics=mets, seq_data=tb)
            mets_im = {k: torch.as_tensor(v).clone() for k, v in mets.items() if k.startswith('imag_')}
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
    assert tape.pos == len(tape.tape), (tape.pos, len(tape.tape))
    flat('metrics_wm.', mets_wm, out); flat('metrics_conn2.', mets_c2, out); flat('metrics_imag.', mets_im, out)
    post_idx = outputs['post']['stoch'].detach().argmax(-1).to(torch.int16)
    prior_idx = outputs['prior']['stoch'].detach().argmax(-1).to(torch.int16)
    out['post_idx'] = post_idx.numpy(); out['prior_idx'] = prior_idx.numpy()
    out['imag_idx'] = cap['seq']['stoch'].argmax(-1).to(torch.int16).numpy()
    out['target_idx'] = cap['target_stoch'].argmax(-1).to(torch.int16).numpy()
    tensors = dict(embed=outputs['embed'], post_logit=outputs['post']['logit'], post_deter=outputs['post']['deter'],
                   prior_logit=outputs['prior']['logit'], like_obs=outputs['likes']['observation'],
                   like_rew=outputs['likes']['reward'], kl=outputs['kl'],
                   imag_feat=cap['seq']['feat'], imag_action=cap['seq']['action'], reward=cap['reward'],
                   lambda_target=cap['lambda_target'])
    for k, v in tensors.items():
        out[f'sum.{k}'] = summarize(v)
        if full_tensors:
            out[f'full.{k}'] = v.detach().numpy()
    for ph, gd in grads.items():
        for n, gten in gd.items():
            out[f'gsum.{ph}.{n}'] = summarize(gten, 4)
            if full_tensors and gten.numel() <= 20000 and ph != 'conn2':
                out[f'grad.{ph}.{n}'] = gten.numpy()
    if not lr_zero:
        for n, pten in ag.state_dict().items():
            out[f'psum.{n}'] = summarize(pten - det[n], 4)
    out['meta'] = np.array([B, T, A, S, K, H, seed, int(lr_zero)])
    out['torch_version'] = np.array(torch.__version__)
    if store_batch:
        flat('batch.', batch, out)
    out['img'] = np.array(img)
    return out


def run_dreamer(B, T, A, over, seed):
    """DreamerAgent.update (update_wm + update_acting_behavior with env_reward), lr = 0."""
    over = dict(over)
    for k in ('model_opt', 'actor_opt', 'critic_opt'):
        over[k] = dict(lr=0.0, wd=0.0)
    ag = rh.make_ref_dreamer(B, T, A=A, **over)
    for d_ in ag._acting_behavior._target_critic.parameters():
        d_.data = d_.data.clone()
    det = detgen.det_state_dict({k: v.shape for k, v in ag.state_dict().items()}, seed)
    ag.load_state_dict(det)
    S, K, H = ag.cfg.rssm.stoch, ag.cfg.rssm.discrete, ag.cfg.imag_horizon
    noise = detgen.iteration_noise(B, T, S, K, A, H, seed=seed)
    tape = []
    for t in range(T):
        tape.append(('exp', noise['wm']['prior_q'][t])); tape.append(('exp', noise['wm']['post_q'][t]))
    tape.append(('normal', noise['imag']['act_eps0']))
    for h in range(H):
        tape.append(('normal', noise['imag']['act_eps'][h])); tape.append(('exp', noise['imag']['step_q'][h]))
    tape = rh.NoiseTape('replay', tape)
    names = {id(p): n for n, p in ag.named_parameters()}
    grads, phase = {}, ['wm']
    orig_clip = torch.nn.utils.clip_grad_norm_

    def clip_capture(params, clip, *a, **k):
        params = list(params)
        grads[phase[0]] = {names[id(p)]: p.grad.detach().clone() for p in params if p.grad is not None}
        return orig_clip(params, clip, *a, **k)
    batch = detgen.det_batch(B, T, A=A, seed=seed)
    tb = {k: v for k, v in rh.to_torch(batch).items() if k != 'clip_video'}
    torch.nn.utils.clip_grad_norm_ = clip_capture
    out = {}
    try:
        with rh.inject_noise(tape):
            state, outputs, mets = ag.update_wm(tb, 0)
            mets_wm = {k: torch.as_tensor(v).clone() for k, v in mets.items()}
            phase[0] = 'actor'
            orig_cl = ag._acting_behavior.critic_loss

            def critic_hook(*a, **k):
                phase[0] = 'critic'
                return orig_cl(*a, **k)
            ag._acting_behavior.critic_loss = critic_hook
            _, mets = ag.update_acting_behavior(state, outputs, {}, tb)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
    assert tape.pos == len(tape.tape), (tape.pos, len(tape.tape))
    flat('metrics_wm.', mets_wm, out); flat('metrics_act.', {k: torch.as_tensor(v) for k, v in mets.items()}, out)
    out['post_idx'] = outputs['post']['stoch'].detach().argmax(-1).to(torch.int16).numpy()
    for ph, gd in grads.items():
        for n, gten in gd.items():
            out[f'gsum.{ph}.{n}'] = summarize(gten, 4)
    out['meta'] = np.array([B, T, A, S, K, H, seed, 1])
    out['torch_version'] = np.array(torch.__version__)
    out['img'] = np.array(64)
    return out


def main():
    kat()
    tiny = detgen.tiny_overrides()
    b = rh.stickman_batch(2, 16, seed=1)
    o = run_iteration(2, 16, 10, tiny, seed=0, lr_zero=True, full_tensors=True, batch=b)
    np.savez_compressed(f'{HERE}/tiny_iter.npz', **o); print('tiny_iter.npz', len(o))
    o = run_iteration(2, 16, 10, tiny, seed=0, lr_zero=False, full_tensors=False, batch=b)
    np.savez_compressed(f'{HERE}/tiny_opt.npz', **o); print('tiny_opt.npz', len(o))
    b = rh.stickman_batch(4, 16, seed=0)
    o = run_iteration(4, 16, 10, {}, seed=0, lr_zero=True, full_tensors=False, batch=b)
    np.savez_compressed(f'{HERE}/c1_full.npz', **o); print('c1_full.npz', len(o))
    # c4-like: 128x128 frames need the 5-layer conv stacks; kitchen A=9; tiny widths;
    # the batch (incl. an is_first inside a window) is regenerated from its seed, not stored
    c4 = dict(tiny); c4['encoder'] = dict(cnn_depth=4, cnn_kernels=[4, 4, 4, 4, 4]); c4['decoder'] = dict(cnn_depth=4, cnn_kernels=[5, 5, 5, 6, 6])
    b = detgen.det_batch(2, 16, A=9, img=128, seed=4)
    o = run_iteration(2, 16, 9, c4, seed=4, lr_zero=True, full_tensors=False, batch=b, img=128, store_batch=False)
    np.savez_compressed(f'{HERE}/c4_tiny.npz', **o); print('c4_tiny.npz', len(o))
    # c3-like: DreamerAgent + dreamer_v3.yaml (walker A=6, T=50 is not a multiple of 8), tiny widths
    o = run_dreamer(2, 18, 6, detgen.dreamer_tiny_overrides(), seed=3)
    np.savez_compressed(f'{HERE}/c3_dreamer_tiny.npz', **o); print('c3_dreamer_tiny.npz', len(o))


if __name__ == '__main__':
    main()
