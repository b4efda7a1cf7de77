"""MI355X-native `agent/dreamer.py`: DreamerAgent, WorldModel, ActorCritic with the reference's
API (mazpie/genrl agent/dreamer.py) on top of the HIP kernels.  Same attribute / method /
metric names; see SURVEY.md §8b for the call contract with train.py."""
import contextlib
from collections import OrderedDict

import numpy as np
import os
import torch
import torch.nn as nn

from . import dreamer_utils as common
from .. import graph, noise, ops, ops_planes, streams, planes
from ..tools.genrl_utils import *          # reward functions resolved through globals(), ref :9


def stop_gradient(x):
    return x.detach()


Module = common.Module


def env_reward(agent, seq):  # ref :16-17
    return agent.wm.heads['reward'](seq['feat']).mean


class DreamerAgent(Module):
    """Acting + training facade over a WorldModel and an ActorCritic (agent/dreamer.py:19-118).
    `train.py` / `collect_data.py` use: act, update_wm, update_acting_behavior, update, report,
    init_meta / update_meta / get_meta_specs."""
    def __init__(self, name, cfg, obs_space, act_spec, **kwargs):
        super().__init__()
        cfg.update(**kwargs)
        self.name, self.cfg, self.obs_space, self.act_spec = name, cfg, obs_space, act_spec
        self.device = cfg.device
        self.act_dim = act_spec.shape[0]
        self._use_amp = cfg.precision == 16
        # precision 16 (the reference wraps its forward passes in fp16 autocast + GradScaler, agent/dreamer.py:38,
        # dreamer_utils.py:889-932): here the MFMA GEMMs round their operands to bf16 and accumulate in fp32, every
        # tensor stays fp32, and no scaler is needed (bf16 keeps fp32's exponent range).  Process-wide switch.
        self._apply_precision()
        self.wm = WorldModel(cfg, obs_space, self.act_dim)
        self.instantiate_acting_behavior()
        self.to(self.device)
        self.requires_grad_(requires_grad=False)       # every update switches on exactly what it trains

    def instantiate_acting_behavior(self):
        self._acting_behavior = ActorCritic(self.cfg, self.act_spec, self.wm.inp_size).to(self.device)

    # ------------------------------------------------------------------ acting (agent/dreamer.py:41-64)
    def act(self, obs, meta, step, eval_mode, state):
        """One environment step: filter the latent with the new observation, then query the actor.
        `state` is (latent, previous action) or None at episode start."""
        if self.cfg.only_random_actions:
            return np.random.uniform(-1, 1, self.act_dim).astype(self.act_spec.dtype), (None, None)
        self._apply_precision()
        batch = {k: torch.as_tensor(np.copy(v), device=self.device).unsqueeze(0) for k, v in obs.items()}
        if state is not None:
            latent, prev_action = state
        else:
            n = len(batch['reward'])
            latent = self.wm.rssm.initial(n)
            prev_action = torch.zeros((n,) + tuple(self.act_spec.shape), device=self.device)
        sample_latent = (not eval_mode) or (not self.cfg.eval_state_mean)
        with torch.no_grad():
            embed = self.wm.encoder(self.wm.preprocess(batch))
            latent, _ = self.wm.rssm.obs_step(latent, prev_action, embed, batch['is_first'], sample_latent)
            policy = self._acting_behavior.actor(self.wm.rssm.get_stoch(latent), latent['deter'])
            action = policy.mean if eval_mode else policy.sample()
        return action.cpu().numpy()[0], (latent, action)

    def _apply_precision(self):
        """The GEMM arithmetic mode is process-wide in the library: every entry point of an agent (re)selects its own, so
        that two agents of different `precision` in one process do not change each other's arithmetic."""
        ops.set_gemm_precision('bf16' if self._use_amp else ops.F32_MODE)
        # precision 16 is ONE arithmetic -- every matrix product rounds both operands to bf16 and accumulates in fp32 (what
        # oracle/genrl_oracle.py restates as `bf16_operands`, tests/test_gpu_iteration.py::test_precision16_vs_the_oracles_bf16_operand_mode) -- so the pre-split fp16-plane
        # products (fp32-grade) are switched off while an agent of that precision runs, and back on for the next fp32 agent
        from .. import planes
        if self._use_amp:
            if planes._amp_saved is None:
                planes._amp_saved = planes.ENABLED
            planes.ENABLED = False
        elif planes._amp_saved is not None:
            planes.ENABLED, planes._amp_saved = planes._amp_saved, None

    # ------------------------------------------------------------------ training entry points
    def update_wm(self, data, step):  # agent/dreamer.py:66-71
        self._apply_precision()
        noise.new_step(next(iter(data.values())).device)      # (one RNG launch per kind for the whole iteration)
        state, outputs, wm_metrics = self.wm.update(data, state=None)
        outputs['is_terminal'] = data['is_terminal']
        return state, outputs, dict(wm_metrics)

    def _posterior_for(self, outputs, data):
        """Posterior states to start imagination from: the world-model update's, or a fresh
        no-grad observe pass over `data`."""
        if outputs is not None:
            return outputs['post'], outputs['is_terminal']
        data = self.wm.preprocess(data)
        with torch.no_grad():
            post, _ = self.wm.rssm.observe(self.wm.encoder(data), data['action'], data['is_first'])
        return post, data['is_terminal']

    def update_acting_behavior(self, state=None, outputs=None, metrics={}, data=None, reward_fn=None):  # :73-92
        if self.cfg.only_random_actions:
            return {}, metrics
        post, is_terminal = self._posterior_for(outputs, data)
        start = {k: stop_gradient(v) for k, v in post.items()}
        fn = reward_fn if reward_fn is not None else globals()[self.cfg.acting_reward_fn]
        metrics.update(self._acting_behavior.update(self.wm, start, is_terminal, lambda seq: fn(self, seq)))
        return start, metrics

    def update(self, data, step):
        state, outputs, metrics = self.update_wm(data, step)
        _, metrics = self.update_acting_behavior(state, outputs, metrics, data)
        return state, metrics

    def report(self, data):  # agent/dreamer.py:99-109
        data = self.wm.preprocess(data)
        videos = {}
        with torch.no_grad():
            for key in self.wm.heads['decoder'].cnn_keys:
                videos['openl_' + key.replace('/', '_')] = self.wm.video_pred(data, key)
            for extra in getattr(self.cfg, 'additional_report_fns', []):
                videos.update(globals()[extra](self, data))
        return videos

    def get_meta_specs(self):
        return tuple()

    def init_meta(self):
        return OrderedDict()

    def update_meta(self, meta, global_step, time_step, finetune=False):
        return meta


_constants = {}


def _constant(shape, value, dev):
    """read-only constant tensor, cached per (shape, value, device)"""
    key = (tuple(shape), float(value), str(dev))
    if key not in _constants:
        _constants[key] = torch.full(tuple(shape), float(value), device=dev)
    return _constants[key]


def _state_planes(seq):
    """((stoch planes, 0), (deter planes, 0)) of an imagined sequence whose rollout produced them, else None"""
    p_ = getattr(seq, 'planes', None)
    return ((p_[0], 0), (p_[1], 0)) if p_ is not None else None


class _ImaginedSeq(dict):
    """WorldModel.imagine's result.  The reference stores seq['feat'] = cat(stoch, deter) eagerly (agent/dreamer.py:272);
    on the GenRL path nothing reads it (the heads take stoch and deter as two operands, video_text_reward goes
    through the connector), and at 17 x 1024 rows the concatenation is a 140 MB copy per update -- so it is built
    the first time somebody asks for it (env_reward does)."""
    def __init__(self, rssm, items):
        super().__init__(items)
        self._rssm = rssm

    def __missing__(self, key):
        if key != 'feat':
            raise KeyError(key)
        value = self._rssm.get_feat(self)
        self[key] = value
        return value


class WorldModel(Module):  # ref :120-321
    def __init__(self, config, obs_space, act_dim):
        super().__init__()
        shapes = {k: tuple(v.shape) for k, v in obs_space.items()}
        self.shapes = shapes
        self.cfg = config
        self.device = config.device
        self.encoder = common.Encoder(shapes, **config.encoder)
        key = self.encoder.cnn_keys[0]
        sizes = common._conv_sizes(shapes[key][-1], config.encoder['cnn_kernels'])
        embed_dim = 2 ** (len(sizes) - 1) * self.encoder._cnn_depth * sizes[-1] ** 2     # ref :129-132
        self.embed_dim = embed_dim
        self.rssm = common.EnsembleRSSM(**config.rssm, action_dim=act_dim, embed_dim=embed_dim, device=self.device)
        self.heads = {}
        self._use_amp = (config.precision == 16)
        self.inp_size = self.rssm.get_feat_size()
        self.decoder_input_fn = getattr(self.rssm, f'get_{config.decoder_inputs}')
        self.decoder_input_size = getattr(self.rssm, f'get_{config.decoder_inputs}_size')()
        self.heads['decoder'] = common.Decoder(shapes, **config.decoder, embed_dim=self.decoder_input_size,
                                               image_dist=config.image_dist)
        self.heads['reward'] = common.MLP(self.inp_size, (1,), **config.reward_head)
        with torch.no_grad():
            for p in self.heads['reward']._out.parameters():
                p.data = p.data * 0
        assert not config.pred_discount, 'discount head is not on the GenRL path (conf/env/dmc_pixels.yaml:6)'
        for name in config.grad_heads:
            assert name in self.heads, name
        self.grad_heads = config.grad_heads
        self.heads = nn.ModuleDict(self.heads)
        self.model_opt = common.Optimizer('model', self.parameters(), **config.model_opt, use_amp=self._use_amp)
        self.e2e_update_fns = {}
        self.detached_update_fns = {}
        self.eval()

    def add_module_to_update(self, name, module, update_fn, detached=False):  # ref :158-164
        self.add_module(name, module)
        (self.detached_update_fns if detached else self.e2e_update_fns)[name] = update_fn
        self.model_opt = common.Optimizer('model', self.parameters(), **self.cfg.model_opt, use_amp=self._use_amp)

    def update(self, data, state=None):  # ref :166-187
        streams.join()
        self.train()
        with common.RequiresGrad(self):
            assert not (getattr(self.cfg, 'freeze_decoder', False) or getattr(self.cfg, 'freeze_post', False)
                        or getattr(self.cfg, 'freeze_model', False)), 'freeze_* modes are off the north-star path'
            model_loss, state, outputs, metrics = self.loss(data, state)
            model_loss, metrics = self.update_additional_e2e_modules(data, outputs, model_loss, metrics)
            # (data parallel: the gradient reduction is started here and completed behind the connector update's
            # forward + backward below -- the connector reads the posterior, not the world-model weights)
            metrics.update(self.model_opt(model_loss, self.parameters(), defer=len(self.detached_update_fns) > 0))
        if len(self.detached_update_fns) > 0:
            detached_loss, metrics = self.update_additional_detached_modules(data, outputs, metrics)
        self.model_opt.flush()
        self.eval()
        return state, outputs, metrics

    def update_additional_detached_modules(self, data, outputs, metrics):  # ref :189-200
        detached_loss = 0
        # cfg.overlap_detached: enqueue these updates on a side stream (genrl_amd/streams.py); they are
        # joined before update_imag_behavior returns / before the next update_wm starts.
        # (only when a behaviour update follows: GenRLAgent.update_imag_behavior joins the side stream; in the
        # pre-training configuration -- imag_reward_fn None -- train.py never calls it, so nothing would order the side
        # stream's reads of the batch before the caller recycles it)
        # Data parallel: only with stream-ordered collectives (RCCL: Optimizer.overlap_under_dp) -- the branch's own
        # reductions then queue behind the world model's on the backend's stream, in the same program order on every rank,
        # and the world-model step that is still pending (its reduction runs beside this branch) is completed by the MAIN
        # stream (flush_pending=False here; WorldModel.update flushes after the fork).
        overlap = (getattr(self.cfg, 'overlap_detached', False)
                   and (common.Optimizer.grad_reduce is None or (common.Optimizer.overlap_under_dp and not graph.cutting()))
                   and getattr(self.cfg, 'imag_reward_fn', None) is not None)
        ctx = streams.fork('detached') if overlap else contextlib.nullcontext()
        with ctx:
            for k in self.detached_update_fns:
                detached_module = getattr(self, k)
                with common.RequiresGrad(detached_module):
                    add_loss, add_metrics = self.detached_update_fns[k](self, k, data, outputs, metrics)
                    metrics.update(add_metrics)
                    opt_metrics = self.model_opt(add_loss, detached_module.parameters(), flush_pending=not overlap)
                    metrics.update({f'{k}_{m}': opt_metrics[m] for m in opt_metrics})
        return detached_loss, metrics

    def update_additional_e2e_modules(self, data, outputs, model_loss, metrics):  # ref :202-208
        for k in self.e2e_update_fns:
            add_loss, add_metrics = self.e2e_update_fns[k](self, k, data, outputs, metrics)
            model_loss = model_loss + add_loss
            metrics.update(add_metrics)
        return model_loss, metrics

    def observe_data(self, data, state=None):  # ref :210-217
        data = self.preprocess(data)
        embed = self.encoder(data)
        post, prior = self.rssm.observe(embed, data['action'], data['is_first'], state)
        kl_loss, kl_value = self.rssm.kl_loss(post, prior, **self.cfg.kl)
        outs = dict(embed=embed, post=post, prior=prior, is_terminal=data['is_terminal'])
        return outs, {'model_kl': kl_value.mean()}

    def loss(self, data, state=None):  # ref :219-252
        data = self.preprocess(data)
        embed = self.encoder(data)
        # cfg.overlap_detached: the prior branch (GRU scan -> prior head -> KL) runs on a side stream
        # beside the decoder / reward-head branch (genrl_amd/streams.py); only possible when the heads
        # do not read `deter` with a gradient path that the KL shares, i.e. the GenRL configuration
        # (not under data parallelism: at the per-rank batch sizes the fork / join idles cost more than the branch hides,
        # DESIGN par.6)
        fork = (getattr(self.cfg, 'overlap_detached', False) and self.rssm.single_obs_posterior
                and self.cfg.decoder_inputs == 'stoch' and self.grad_heads == ['decoder']
                and common.Optimizer.grad_reduce is None)
        self.rssm.fork_prior = fork
        post, prior = self.rssm.observe(embed, data['action'], data['is_first'], state)
        self.rssm.fork_prior = False
        likes = {}
        losses = {'kl': None}
        joined = not fork
        for name, head in self.heads.items():
            grad_head = (name in self.grad_heads)
            if name == 'decoder':
                inp = self.decoder_input_fn(post)            # posterior stoch: produced on the main stream
                inp = inp if grad_head else stop_gradient(inp)
                out = head(inp)
            else:       # MLP heads consume feat = [stoch, deter] without the concatenation
                if not joined:                               # deter comes from the side-stream scan
                    streams.join('scan'); joined = True
                s, d = self.rssm.get_stoch(post), post['deter']
                if not grad_head:
                    s, d = stop_gradient(s), stop_gradient(d)
                out = head(s, d)
            dists = out if isinstance(out, dict) else {name: out}
            for key, dist in dists.items():
                like = dist.log_prob(data[key])
                likes[key] = like
                losses[key] = ops.wmean(like, None, -1.0)        # -like.mean() as one node
        if not joined:
            streams.join('scan')
        kl_loss, kl_value = self.rssm.kl_loss(post, prior, **self.cfg.kl)
        assert len(kl_loss.shape) == 0 or (len(kl_loss.shape) == 1 and kl_loss.shape[0] == 1), kl_loss.shape
        losses['kl'] = kl_loss
        feat = self.rssm.get_feat(post)
        scaled = lambda k, v: v if self.cfg.loss_scales.get(k, 1.0) == 1.0 else self.cfg.loss_scales[k] * v
        model_loss = sum(scaled(k, v) for k, v in losses.items())
        outs = dict(embed=embed, feat=feat, post=post, prior=prior, likes=likes, kl=kl_value)
        metrics = {f'{name}_loss': value for name, value in losses.items()}
        metrics['model_kl'] = ops.wmean(kl_value.detach(), None, 1.0)
        metrics['prior_ent'] = ops.wmean(self.rssm.get_dist(prior).entropy(), None, 1.0)
        metrics['post_ent'] = ops.wmean(self.rssm.get_dist(post).entropy(), None, 1.0)
        last_state = {k: v[:, -1] for k, v in post.items()}
        return model_loss, last_state, outs, metrics

    def imagine(self, policy, start, is_terminal, horizon, task_cond=None, eval_policy=False):  # ref :254-287
        assert task_cond is None
        flatten = lambda x: x.reshape([-1] + list(x.shape[2:]))
        start = {k: flatten(v) for k, v in start.items()}
        N = start['deter'].shape[0]
        dev = start['deter'].device
        A = policy._out._out.out_features
        eps = noise.draw('normal', 'imag.act_eps', (horizon, N, A), dev) if not eval_policy else None
        q = noise.draw('exp', 'imag.step_q', (horizon, N * self.rssm._stoch, self.rssm._discrete), dev)
        rssm = self.rssm
        seq = {k: [v] for k, v in start.items()}
        seq['action'] = [torch.zeros(N, A, device=dev)]
        # the two heads of DistLayer('normal') (mean, std) as ONE product: weights stacked once per rollout
        head_w = torch.cat([policy._out._out.weight, policy._out._std.weight], 0)
        head_b = torch.cat([policy._out._out.bias, policy._out._std.bias], 0)
        raws = []
        # training rollouts: the policy's H backward passes are batched into one over all H*N rows
        tape = None
        if torch.is_grad_enabled() and head_w.requires_grad and horizon > 1 and policy._norm != 'none':
            layers = [(getattr(policy, f'dense{i}').weight, getattr(policy, f'dense{i}').bias,
                       getattr(policy, f'norm{i}')._layer.weight, getattr(policy, f'norm{i}')._layer.bias,
                       getattr(policy, f'norm{i}')._layer.eps) for i in range(policy._layers)]
            # (plane operands pay from ~512 rollout rows up: below, every product is launch-latency bound either way and
            # the plane writes only add traffic -- measured 14.2 vs 13.7 ms/step at 4 sequences per GPU)
            use_planes = planes.ENABLED and N >= ops_planes.min_rows()
            tape = (ops_planes.ActorTapePlanes if use_planes else ops.ActorTape)(horizon, N, layers, head_w, head_b, dev)
            tape.head_leaves = (policy._out._out.weight, policy._out._out.bias, policy._out._std.weight, policy._out._std.bias)
        fused = tape is not None and not eval_policy and set(start) == {'stoch', 'deter', 'logit'}
        if fused:
            # the whole H-step loop as one autograd node (ops._Rollout): dynamics dgrad chain + policy tape
            inl, inn = rssm._img_in[0], rssm._img_in[1]._layer
            outl, outn = rssm._ensemble_img_out[0][0], rssm._ensemble_img_out[0][1]._layer
            dist = rssm._ensemble_img_dist[0]
            spec = ops.RolloutSpec(tape, inl.weight, inl.bias, inn.weight, inn.bias, inn.eps,
                                   rssm._cell._layer.weight, rssm._cell._norm.weight, rssm._cell._norm.bias,
                                   outl.weight, outl.bias, outn.weight, outn.bias, outn.eps, dist.weight, dist.bias,
                                   rssm._stoch, rssm._discrete, policy._out._min_std, policy._out._max_std)
            roll = ops_planes.imagine_rollout if isinstance(tape, ops_planes.ActorTapePlanes) else ops.imagine_rollout
            st, de, lg, ac, raw_all = roll(start['stoch'], start['deter'], start['logit'], eps, q, spec)
            seq = {'stoch': st, 'deter': de, 'logit': lg, 'action': ac}
            self._last_actor_raw = raw_all
            state_planes = getattr(tape, 'state_planes', None)
        else:
            for h in range(horizon):
                stoch, deter = seq['stoch'][-1], seq['deter'][-1]
                s_flat = stoch.reshape(N, -1)
                if tape is not None:
                    raw = tape.step(h, stop_gradient(s_flat), stop_gradient(deter))
                else:
                    raw = ops.linear(policy.trunk(stop_gradient(s_flat), stop_gradient(deter)), head_w, head_b)
                raws.append(raw)
                if eval_policy:
                    action = ops.actor_mean_std(raw, policy._out._min_std, policy._out._max_std)[0]
                else:
                    action = ops.actor_sample(raw, eps[h], policy._out._min_std, policy._out._max_std)
                x = common._dense_ln_silu(s_flat, rssm._img_in[0], rssm._img_in[1], action)
                deter = ops.gru_step(x, deter, rssm._cell._layer.weight, rssm._cell._norm.weight, rssm._cell._norm.bias)
                logit = rssm._prior_logits(deter)
                stoch = ops.onehot_sample(logit, q[h])
                for key, value in dict(stoch=stoch, deter=deter, logit=logit, action=action).items():
                    seq[key].append(value)
            seq = {k: torch.stack(v, 0) for k, v in seq.items()}
            if tape is not None:     # layer-0 inputs of all steps = the stacked rollout states (no copies)
                tape.inputs = (seq['stoch'].detach().reshape(horizon + 1, N, -1), seq['deter'].detach())
            # policy outputs at states 0..H-1 — exactly what ActorCritic.actor_loss re-evaluates for its
            # entropy metric (agent/dreamer.py:397: actor(sg(feat[:-2]))): kept to avoid a second forward
            self._last_actor_raw = torch.stack(raws, 0)          # (H, N, 2A), attached to the actor's graph
        seq = _ImaginedSeq(rssm, seq)                  # 'feat' = cat(stoch, deter) (ref :272) on first access
        seq.planes = locals().get('state_planes')          # (planes of stoch / deter, rows h*N + n, from the fused rollout)
        # no discount head (conf/env/dmc_pixels.yaml:6): discount = gamma everywhere and weight = cumprod(ones) = 1
        # (ref :274-286) -- constants, built once per shape instead of five launches per update
        shape = tuple(seq['deter'].shape[:-1]) + (1,)
        seq['discount'] = _constant(shape, float(self.cfg.discount), dev)
        seq['weight'] = _constant(shape, 1.0, dev)
        seq.unit_weight = True
        return seq

    def preprocess(self, obs):  # ref :289-305; uint8 frames stay uint8 (x/255-0.5 is fused downstream)
        obs = dict(obs)
        assert self.cfg.clip_rewards == 'identity'
        obs['discount'] = (1.0 - obs['is_terminal'].float())
        if 'reward' in obs and len(obs['discount'].shape) < len(obs['reward'].shape):
            obs['discount'] = obs['discount'].unsqueeze(-1)
        return obs

    def video_pred(self, data, key, nvid=8):  # ref :307-321
        decoder = self.heads['decoder']
        truth = data[key][:nvid].float() / 255.0
        embed = self.encoder(data)
        states, _ = self.rssm.observe(embed[:nvid, :5], data['action'][:nvid, :5], data['is_first'][:nvid, :5])
        recon = decoder(self.decoder_input_fn(states))[key].mean[:nvid]
        init = {k: v[:, -1] for k, v in states.items()}
        prior = self.rssm.imagine(data['action'][:nvid, 5:], init)
        prior_recon = decoder(self.decoder_input_fn(prior))[key].mean
        model = torch.clip(torch.cat([recon[:, :5] + 0.5, prior_recon + 0.5], 1), 0, 1)
        error = (model - truth + 1) / 2
        return torch.cat([truth, model, error], 3)


class ActorCritic(Module):  # ref :323-462
    def __init__(self, config, act_spec, feat_size, name=''):
        super().__init__()
        self.name = name
        self.cfg = config
        self.act_spec = act_spec
        self._use_amp = (config.precision == 16)
        self.device = config.device
        assert not getattr(self.cfg, 'discrete_actions', False)
        self.actor_grad = getattr(self.cfg, f'{self.name}_actor_grad'.strip('_'))
        assert self.actor_grad == 'dynamics', 'GenRL trains the actor through the dynamics'
        self.actor = common.MLP(feat_size, act_spec.shape[0], **self.cfg.actor)
        self.critic = common.MLP(feat_size, (1,), **self.cfg.critic)
        assert self.cfg.slow_target and self.cfg.reward_ema
        self._target_critic = common.MLP(feat_size, (1,), **self.cfg.critic)
        self._updates = 0
        self.actor_opt = common.Optimizer('actor', self.actor.parameters(), **self.cfg.actor_opt, use_amp=self._use_amp)
        self.critic_opt = common.Optimizer('critic', self.critic.parameters(), **self.cfg.critic_opt, use_amp=self._use_amp)
        self.register_buffer('ema_vals', torch.zeros((2,)).to(self.device))
        self.reward_ema = common.RewardEMA(device=self.device)
        self.rewnorm = common.StreamNorm(momentum=1, scale=1.0, device=self.device)
        with torch.no_grad():
            for p in self.critic._out.parameters():
                p.data = p.data * 0
            for s, d in zip(self.critic.parameters(), self._target_critic.parameters()):
                d.data = s.data.clone()
        # the slow critic is never trained: no wgrad for it (drops the reference's waste)
        self._target_critic.requires_grad_(False)

    def update(self, world_model, start, is_terminal, reward_fn):  # ref :366-390
        metrics = {}
        hor = self.cfg.imag_horizon
        self._target_critic.requires_grad_(False)
        # the critic update beside the actor's backward, on its own stream: measured per config (profiles/r05_fork_ab.txt) -- it pays only
        # where the connector updates are running on THEIR side stream at the same time (c2 23.5 -> 23.4 ms, c4 44.2 -> 43.9) and costs
        # where nothing else runs beside (c3 38.9 -> 40.0 ms, 13.9 -> 15.2 at 8 sequences; c5 9.4 -> 10.1): forked only in the first
        # case (GENRL_FORK_CRITIC=1 / 0: always / never)
        fc = os.environ.get('GENRL_FORK_CRITIC', 'auto')
        overlap = (getattr(self.cfg, 'overlap_detached', False) and common.Optimizer.grad_reduce is None
                   and fc != '0' and (fc == '1' or streams.pending('detached')))
        with common.RequiresGrad(self.actor):
            seq = world_model.imagine(self.actor, start, is_terminal, hor)
            self._rollout_actor_raw = getattr(world_model, '_last_actor_raw', None)
            reward = reward_fn(seq)
            seq['reward'], mets1 = self.rewnorm(reward)
            mets1 = {f'reward_{k}': v for k, v in mets1.items()}
            target, mets2, baseline = self.target(seq)
            actor_loss, mets3 = self.actor_loss(seq, target, baseline)
            seq_d = _ImaginedSeq(getattr(seq, '_rssm', None), {k: stop_gradient(v) for k, v in seq.items()})
            seq_d.unit_weight = getattr(seq, 'unit_weight', False)
            seq_d.planes = getattr(seq, 'planes', None)
            target_d = stop_gradient(target)

            def critic_step():
                with common.RequiresGrad(self.critic):
                    critic_loss, mets4_ = self.critic_loss(seq_d, target_d)
                    return self.critic_opt(critic_loss, self.critic.parameters()), mets4_
            if overlap:
                # the critic regression only needs the (detached) rollout and targets: run it on a side
                # stream concurrently with the actor's BPTT — two GEMM streams fill each CU with two
                # independent workgroups.  Joined below, before the slow-target copy.
                with streams.fork('critic'):
                    cm, mets4 = critic_step()
            # (data parallel: the actor's reduction runs beside the critic update below)
            metrics.update(self.actor_opt(actor_loss, self.actor.parameters(), defer=not overlap))
        if overlap:
            streams.join('critic')
        else:
            cm, mets4 = critic_step()
        self.actor_opt.flush()
        metrics.update(cm)
        metrics.update(**mets1, **mets2, **mets3, **mets4)
        if not getattr(self, '_defer_slow_target', False):     # hipGraph mode: the driver calls it per replay
            self.update_slow_target()
        return {f'{self.name}_{k}'.strip('_'): v for k, v in metrics.items()}

    def actor_loss(self, seq, target, baseline):  # ref :392-429 (actor_grad 'dynamics')
        metrics = {}
        offset, scale = self.reward_ema(target, self.ema_vals)
        ent_scale = self.cfg.actor_ent
        weight = stop_gradient(seq['weight'])
        fused = ent_scale == 0 and getattr(self.reward_ema, 'last', None) is not None
        if fused:
            # normalisation, weighted mean, sign and the two 'normed_target' statistics: one launch forward, one backward
            actor_loss, st = ops.actor_objective(target, None if getattr(seq, 'unit_weight', False) else weight[:-2],
                                                 self.reward_ema.last)
            metrics['normed_target_mean'], metrics['normed_target_std'] = st[0], st[1]
            metrics['reward_ema_005'], metrics['reward_ema_095'] = self.ema_vals[0], self.ema_vals[1]
        else:
            normed_target = (target - offset) / scale
            metrics['normed_target_mean'] = normed_target.mean()
            metrics['normed_target_std'] = normed_target.std()
            metrics['reward_ema_005'] = self.ema_vals[0].clone()
            metrics['reward_ema_095'] = self.ema_vals[1].clone()
            objective = normed_target[1:]
        n_pol = seq['stoch'].shape[0] - 2
        raw = getattr(self, '_rollout_actor_raw', None)
        if raw is None or raw.shape[0] < n_pol:       # rollout not produced by WorldModel.imagine: re-evaluate
            s, d = stop_gradient(seq['stoch'][:-2]), stop_gradient(seq['deter'][:-2])
            raw = self.actor._out.raw(self.actor.trunk(s.reshape(list(s.shape[:-2]) + [-1]), d))
        # same weights, same inputs as the rollout's own policy evaluations (the reference re-runs the
        # actor on sg(feat[:-2]), ref :397): their outputs - and graph - are reused
        raw = raw[:n_pol]
        A = raw.shape[-1] // 2
        mn, mx = self.actor._out._min_std, self.actor._out._max_std
        if ent_scale != 0:
            std = (mx - mn) * torch.sigmoid(raw[..., A:] + 2.0) + mn
            ent = (0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std)).sum(-1)[:, :, None]
            objective = objective + ent_scale * ent
            metrics['actor_ent'] = ent.detach().mean()
        else:                           # a metric only: no backward through it (its scale is 0); one launch
            metrics['actor_ent'] = ops.normal_entropy_mean(raw, mn, mx)
        metrics['actor_ent_scale'] = ent_scale
        if not fused:
            actor_loss = -(weight[:-2] * objective).mean()
        return actor_loss, metrics

    def critic_loss(self, seq, target):  # ref :431-438
        s, d = seq['stoch'][:-1], seq['deter'][:-1]
        dist = self.critic(s.reshape(list(s.shape[:-2]) + [-1]), d, planes=_state_planes(seq))
        target = stop_gradient(target)
        weight = stop_gradient(seq['weight'])
        # -(log_prob * weight).mean() as one node (ops.wmean)
        critic_loss = ops.wmean(dist.log_prob(target), None if getattr(seq, 'unit_weight', False) else weight[:-1].squeeze(-1), -1.0)
        with torch.no_grad():
            metrics = {'critic': ops.wmean(dist.mean, None, 1.0)}
        return critic_loss, metrics

    def target(self, seq):  # ref :440-453
        reward, disc = seq['reward'], seq['discount']
        s = seq['stoch']
        value = self._target_critic(s.reshape(list(s.shape[:-2]) + [-1]), seq['deter'], planes=_state_planes(seq)).mean
        # lambda_return(reward[:-1], value[:-1], bootstrap=value[-1]) (ref :446-449) on the unsliced tensors: the
        # slices + re-concatenation are three copies forward and three backward otherwise
        assert not isinstance(self.cfg.discount, torch.Tensor)
        target = ops.lambda_return(reward, value, float(self.cfg.discount), float(self.cfg.discount_lambda))
        metrics = {'critic_slow': ops.wmean(value.detach(), None, 1.0), 'critic_target': ops.wmean(target.detach(), None, 1.0)}
        return target, metrics, value[:-1]

    def update_slow_target(self):  # ref :455-462
        if self._updates % self.cfg.slow_target_update == 0:
            mix = 1.0 if self._updates == 0 else float(self.cfg.slow_target_fraction)
            with torch.no_grad():
                for s, d in zip(self.critic.parameters(), self._target_critic.parameters()):
                    d.data.copy_(mix * s.data + (1 - mix) * d.data)
            planes.refresh(list(self._target_critic.parameters()))      # (cached weight planes of the slow critic)
        self._updates += 1
