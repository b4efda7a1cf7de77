"""Oracle driver: one train.py iteration (train.py:273-340 with train_world_model=train_connector=
True, imag_reward_fn=video_text_reward) composed from oracle/genrl_oracle.py.  TEST
INFRASTRUCTURE ONLY (see genrl_oracle.py header)."""
import torch
from . import genrl_oracle as O


def _leafs(p, names):
    q = dict(p)
    for n in names:
        q[n] = p[n].detach().clone().requires_grad_(True)
    return q


def _grads(loss, q, names):
    gs = torch.autograd.grad(loss, [q[n] for n in names], allow_unused=True)
    return {n: g for n, g in zip(names, gs) if g is not None}


def group_names(p):
    conn = [n for n in p if n.startswith('wm.connector.')]
    wm = [n for n in p if n.startswith('wm.') and not n.startswith('wm.connector.')]
    actor = [n for n in p if n.startswith('_imag_behavior.actor.')]
    critic = [n for n in p if n.startswith('_imag_behavior.critic.')]
    return dict(wm=wm, conn=conn, actor=actor, critic=critic)


def run_iteration(p, cfg, batch, noise, text_feat, opt_state=None, apply_updates=True,
                  ema_vals=None, target_cache=None):
    """Returns dict(metrics, grads{phase}, outs, seq, reward, lambda_target, p (updated), ...).
    batch values are CPU torch tensors; p is a dict name->tensor (not modified in place)."""
    p = dict(p)
    g = group_names(p)
    opt_state = opt_state if opt_state is not None else {k: {} for k in ('model', 'actor', 'critic')}
    res = dict(grads={}, metrics={})
    B, T = batch['action'].shape[:2]

    # ---- world model step (agent/dreamer.py:166-187)
    q = _leafs(p, g['wm'])
    loss, outs, mets = O.wm_loss(q, cfg, batch, noise['wm'])
    gr = _grads(loss, q, g['wm'])
    res['grads']['wm'] = gr
    res['metrics'].update({k: v.detach() for k, v in mets.items()})
    res['metrics']['model_loss'] = loss.detach()
    res['outs'] = outs
    if apply_updates:
        norm = O.optimizer_step(p, gr, opt_state['model'], decay_only=g['conn'], **cfg.model_opt)
    else:
        norm = O.global_grad_norm(list(gr.values()))
    res['metrics']['model_grad_norm'] = norm
    post = {k: v.detach() for k, v in outs['post'].items()}

    # ---- connector, twice (agent/dreamer.py:184-185 + train.py:279-280)
    for i in (1, 2):
        q = _leafs(p, g['conn'])
        loss, mets = O.connector_loss(q, cfg, batch['clip_video'], post, noise[f'conn{i}'])
        gr = _grads(loss, q, g['conn'])
        res['grads'][f'conn{i}'] = gr
        if apply_updates:
            norm = O.optimizer_step(p, gr, opt_state['model'], **cfg.model_opt)
        else:
            norm = O.global_grad_norm(list(gr.values()))
        res['metrics'].update({k: v.detach() for k, v in mets.items()})
        res['metrics']['connector_model_loss'] = loss.detach()
        res['metrics']['connector_model_grad_norm'] = norm

    # ---- imagination + actor-critic (agent/genrl.py:108-124, agent/dreamer.py:366-390)
    q = _leafs(p, g['actor'] + g['critic'])
    seq = O.imagine(q, cfg, post, noise['imag'])
    if target_cache is None:
        with torch.no_grad():
            target_cache = O.video_imagine_target(p, cfg, text_feat, B * T, cfg.horizon + 1,
                                                  noise['imag']['target_init_q'])
    reward, ts_idx = O.video_text_reward(q, cfg, seq['stoch'], target_cache['stoch'])
    ema_vals = ema_vals if ema_vals is not None else p.get('_imag_behavior.ema_vals', torch.zeros(2))
    actor_loss, critic_loss, lam_t, mets, new_ema = O.actor_critic_losses(q, cfg, seq, reward, ema_vals)
    ga = _grads(actor_loss, q, g['actor'])
    gc = _grads(critic_loss, q, g['critic'])
    res['grads']['actor'], res['grads']['critic'] = ga, gc
    if apply_updates:
        na = O.optimizer_step(p, ga, opt_state['actor'], **cfg.actor_opt)
        nc = O.optimizer_step(p, gc, opt_state['critic'], **cfg.critic_opt)
    else:
        na, nc = O.global_grad_norm(list(ga.values())), O.global_grad_norm(list(gc.values()))
    im = dict(actor_loss=actor_loss.detach(), actor_grad_norm=na, critic_loss=critic_loss.detach(),
              critic_grad_norm=nc, **O.stream_norm_metrics(reward.detach()),
              **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in mets.items()})
    res['metrics'].update({f'imag_{k}': v for k, v in im.items()})
    p['_imag_behavior.ema_vals'] = new_ema.detach()
    res.update(seq={k: v.detach() for k, v in seq.items()}, reward=reward.detach(),
               lambda_target=lam_t.detach(), target_cache=target_cache, ts_idx=ts_idx,
               p=p, opt_state=opt_state)
    return res


def run_dreamer_iteration(p, cfg, batch, noise, apply_updates=False):
    """DreamerAgent.update (agent/dreamer.py:94-97): world-model step, then the acting behaviour
    trained in imagination on the reward head's predictions (acting_reward_fn = env_reward)."""
    p = dict(p)
    wm = [n for n in p if n.startswith('wm.')]
    actor = [n for n in p if n.startswith('_acting_behavior.actor.')]
    critic = [n for n in p if n.startswith('_acting_behavior.critic.')]
    res = dict(grads={}, metrics={})
    q = _leafs(p, wm)
    loss, outs, mets = O.wm_loss(q, cfg, batch, noise['wm'])
    res['grads']['wm'] = _grads(loss, q, wm)
    res['metrics'].update({k: v.detach() for k, v in mets.items()})
    res['metrics']['model_loss'] = loss.detach()
    res['metrics']['model_grad_norm'] = O.global_grad_norm(list(res['grads']['wm'].values()))
    res['outs'] = outs
    assert not apply_updates
    post = {k: v.detach() for k, v in outs['post'].items()}
    q = _leafs(p, actor + critic)
    seq = O.imagine(q, cfg, post, noise['imag'], actor_prefix='_acting_behavior.actor.')
    reward = O.env_reward(q, cfg, seq)
    ema = p.get('_acting_behavior.ema_vals', torch.zeros(2))
    al, cl, lam_t, mets, new_ema = O.actor_critic_losses(q, cfg, seq, reward, ema, prefix='_acting_behavior.')
    ga, gc = _grads(al, q, actor), _grads(cl, q, critic)
    res['grads']['actor'], res['grads']['critic'] = ga, gc
    im = dict(actor_loss=al.detach(), actor_grad_norm=O.global_grad_norm(list(ga.values())), critic_loss=cl.detach(),
              critic_grad_norm=O.global_grad_norm(list(gc.values())), **O.stream_norm_metrics(reward.detach()),
              **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in mets.items()})
    res['metrics'].update(im)
    res.update(seq={k: v.detach() for k, v in seq.items()}, reward=reward.detach(), lambda_target=lam_t.detach())
    return res
