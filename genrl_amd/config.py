"""Default hyper-parameters of the GenRL hot path as one attribute-dict, for callers that do not
go through Hydra (bench.py, tests, smoke).  Values restate conf/defaults/genrl.yaml,
conf/env/dmc_pixels.yaml and agent/genrl.yaml of the reference (config data);
with train.py the reference's own YAML files are used instead and this module is not needed."""
import numpy as np


class AttrDict(dict):
    """OmegaConf stand-in: attribute access, AttributeError on missing keys (the agent code relies
    on getattr(cfg, key, default))."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _ad(x):
    if isinstance(x, dict):
        return AttrDict({k: _ad(v) for k, v in x.items()})
    return x


class Spec:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), dtype


def default_cfg(batch_size=32, batch_length=32, device='cuda', task='stickman_walk', **over):
    rssm = dict(ensemble=1, hidden=1024, deter=1024, stoch=32, discrete=32, norm='layer', std_act='softplus', min_std=0.1)
    c = dict(
        img_size=64,
        rssm=dict(rssm, single_obs_posterior=True),
        reward_head=dict(layers=4, units=1024, norm='layer', dist='twohot'),
        kl=dict(free=1.0, forward=False, balance=0.85, free_avg=False),
        loss_scales=dict(kl=0.6, reward=1.0, discount=1.0, proprio=1.0),
        model_opt=dict(opt='adam', lr=1e-4, eps=1e-8, clip=1000, wd=1e-6),
        decoder_inputs='stoch', image_dist='mse',
        actor=dict(layers=4, units=1024, norm='layer', dist='normal', min_std=0.1),
        critic=dict(layers=4, units=1024, norm='layer', dist='twohot'),
        actor_opt=dict(opt='adam', lr=3e-5, eps=1e-5, clip=100, wd=1e-6),
        critic_opt=dict(opt='adam', lr=3e-5, eps=1e-5, clip=100, wd=1e-6),
        discount=0.99, discount_lambda=0.95, slow_target=True, slow_target_update=100, slow_target_fraction=1,
        slow_baseline=True, reward_ema=True, acting_reward_fn='env_reward', clip_rewards='identity',
        batch_size=batch_size, batch_length=batch_length, imag_horizon=16, eval_state_mean=False,
        precision=32, only_random_actions=False,
        # conf/env/dmc_pixels.yaml
        encoder=dict(mlp_keys='$^', cnn_keys='observation', norm='layer', cnn_depth=48, cnn_kernels=[4, 4, 4, 4],
                     mlp_layers=[400, 400, 400, 400]),
        decoder=dict(mlp_keys='$^', cnn_keys='observation', norm='layer', cnn_depth=48, cnn_kernels=[5, 5, 6, 6],
                     mlp_layers=[400, 400, 400, 400]),
        pred_discount=False, imag_actor_grad='dynamics', actor_grad='dynamics',
        # agent/genrl.yaml
        grad_heads=['decoder'], reward_norm=dict(momentum=1.0, scale=1.0, eps=1e-8), actor_ent=0,
        clip_add_noise=0.0, clip_lafite_noise=0.5,
        connector=dict(token_dropout=0, loss_scale=1, denoising_ae=True, detached_post=True, temporal_embeds=False,
                       rescale_embeds=True),
        connector_rssm=dict(rssm, single_obs_posterior=False, learn_initial=True),
        connector_kl=dict(free=0.0, forward=True, balance=0.8, free_avg=False),
        imag_reward_fn='video_text_reward', imag_reward_norm=dict(momentum=1.0, scale=1.0, eps=1e-8),
        imag_reward_args=dict(score_fn='max_cosine', sample_for_target=False, align_initial=False,
                              weighted_align=False, align_sequence=True, skip_first_target=True),
        device=device, task=task, viclip_encode=True,
    )
    cfg = _ad(c)
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(v)
        else:
            cfg[k] = _ad(v)
    return cfg


def make_agent(cfg, act_dim=10, img=64, clip_dim=512):
    """Build GenRLAgent the way train.py does (train.py:32-37,62)."""
    from .agent.genrl import GenRLAgent
    obs = dict(observation=Spec((3, img, img), np.uint8), is_first=Spec((), bool), is_last=Spec((), bool),
               is_terminal=Spec((), bool), clip_video=Spec((clip_dim,), np.float32))
    return GenRLAgent(name='genrl', cfg=cfg, obs_space=obs, act_spec=Spec((act_dim,), np.float32))


def tiny_overrides():
    r = dict(hidden=32, deter=32, stoch=4, discrete=4)
    return dict(rssm=r, connector_rssm=r, reward_head=dict(units=32), actor=dict(units=32), critic=dict(units=32),
                encoder=dict(cnn_depth=4), decoder=dict(cnn_depth=4))


def dreamer_cfg(batch_size=64, batch_length=50, device='cuda', task='walker_walk', **over):
    """conf/defaults/dreamer_v3.yaml + conf/env/dmc_pixels.yaml + agent/dreamer.yaml (BASELINE
    configs[2]: DreamerAgent, walker A=6)."""
    rssm = dict(ensemble=1, hidden=512, deter=512, stoch=32, discrete=32, norm='layer', std_act='softplus', min_std=0.1,
                single_obs_posterior=False)
    cfg = default_cfg(batch_size, batch_length, device, task)
    for k in ('connector', 'connector_rssm', 'connector_kl', 'imag_reward_fn', 'imag_reward_norm', 'imag_reward_args',
              'clip_add_noise', 'clip_lafite_noise', 'viclip_encode', 'imag_actor_grad'):
        cfg.pop(k, None)
    cfg.update(_ad(dict(rssm=rssm, reward_head=dict(layers=4, units=512, norm='layer', dist='twohot'),
                        decoder_inputs='feat', actor=dict(layers=4, units=512, norm='layer', dist='normal', min_std=0.1),
                        critic=dict(layers=4, units=512, norm='layer', dist='twohot'), imag_horizon=15,
                        grad_heads=['decoder', 'reward'], actor_ent=3e-4)))
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(v)
        else:
            cfg[k] = _ad(v)
    return cfg


def make_dreamer_agent(cfg, act_dim=6, img=64):
    from .agent.dreamer import DreamerAgent
    obs = dict(observation=Spec((3, img, img), np.uint8), is_first=Spec((), bool), is_last=Spec((), bool),
               is_terminal=Spec((), bool))
    return DreamerAgent(name='dreamer', cfg=cfg, obs_space=obs, act_spec=Spec((act_dim,), np.float32))


def dreamer_tiny_overrides():
    r = dict(hidden=32, deter=32, stoch=4, discrete=4)
    return dict(rssm=r, reward_head=dict(units=32), actor=dict(units=32), critic=dict(units=32),
                encoder=dict(cnn_depth=4), decoder=dict(cnn_depth=4))
