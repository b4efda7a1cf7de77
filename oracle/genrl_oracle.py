"""CPU oracle for the GenRL world-model + imagination hot path.

TEST INFRASTRUCTURE ONLY.  This file is a restatement, in plain PyTorch-CPU fp32 with *explicit
noise inputs*, of the arithmetic the reference (mazpie/genrl) performs on the path named by
BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it; the product path (`genrl_amd/`) never does and fails loudly if
its HIP library is missing.

Parity status: PINNED against the reference itself.  `tests/golden/make_golden.py` imports the
reference from /root/reference in the authoring container, runs it with recorded noise and stores
inputs/outputs/gradients in `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file
against those vectors.  The reference's own test-suite holds no vectors for this path
(SURVEY.md §4), so those fixtures are the pin.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Parameters are a flat dict keyed by the reference's `state_dict()` names (SURVEY.md §8a weight
contract), e.g. 'wm.rssm._cell._layer.weight'.
"""
import math
from types import SimpleNamespace

import contextlib

import torch
import torch.nn.functional as _TF


# ----------------------------------------------------------------------------- matrix products
# Every matrix product of the path goes through F.linear / F.conv2d / F.conv_transpose2d below.  Default: torch's fp32
# functions, i.e. the reference's `precision: 32` arithmetic (this is what the golden vectors pin).
# `with bf16_operands():` restates the PRODUCT's `precision: 16` mode (genrl_amd DESIGN 5c) instead: every product -- forward,
# input gradient and weight gradient alike -- rounds BOTH operands to bf16 (nearest even) and accumulates in fp32; everything
# else (LayerNorm, softmax, losses, bias gradients, the optimiser) stays fp32.  That mode is ORACLE-pinned only: the reference's
# own precision-16 path (fp16 autocast + GradScaler, agent/dreamer_utils.py:889-932) runs on CUDA alone and could not be
# recorded, so no fixture of it exists -- "parity unpinned against the reference" for that row.
_BF16_OPERANDS = False
_BF16_ACC64 = False        # (tests only: accumulate the rounded-operand products in float64 -- a second summation order of the SAME arithmetic,
#                            used to measure how far two correct implementations of this mode drift apart: its rounding noise floor)


def _r(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundedProduct(torch.autograd.Function):
    """y = fn(r(x), r(w)) + b;  backward: the gradients of the same function at (r(x), r(w)) for the ROUNDED output gradient
    r(dy) -- dx = r(dy) * r(w), dw = r(dy)^T * r(x) -- and db = sum of the unrounded dy"""
    @staticmethod
    def forward(ctx, x, w, b, kind):
        ctx.kind = kind
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return _RoundedProduct._fn(kind)(_r(x), _r(w), b)

    @staticmethod
    def _fn(kind):
        base = {'linear': lambda x, w, b: _TF.linear(x, w, b), 'conv': lambda x, w, b: _TF.conv2d(x, w, b, stride=2),
                'convT': lambda x, w, b: _TF.conv_transpose2d(x, w, b, stride=2)}[kind]
        if not _BF16_ACC64:
            return base
        return lambda x, w, b: base(x.double(), w.double(), None if b is None else b.double()).float()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        with torch.enable_grad():
            xr, wr = _r(x).requires_grad_(True), _r(w).requires_grad_(True)
            y = _RoundedProduct._fn(ctx.kind)(xr, wr, None)
            gx, gw = torch.autograd.grad(y, (xr, wr), _r(dy))
        db = None
        if ctx.has_b:
            db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.kind == 'linear' else dy.sum((0, 2, 3))
        return gx, gw, db, None


class _Products:
    """torch.nn.functional with the three matrix products switchable to bf16-rounded operands"""
    def __getattr__(self, name):
        return getattr(_TF, name)

    @staticmethod
    def linear(x, w, b=None):
        return _RoundedProduct.apply(x, w, b, 'linear') if _BF16_OPERANDS else _TF.linear(x, w, b)

    @staticmethod
    def conv2d(x, w, b=None, stride=1):
        assert stride == 2
        return _RoundedProduct.apply(x, w, b, 'conv') if _BF16_OPERANDS else _TF.conv2d(x, w, b, stride=2)

    @staticmethod
    def conv_transpose2d(x, w, b=None, stride=1):
        assert stride == 2
        return _RoundedProduct.apply(x, w, b, 'convT') if _BF16_OPERANDS else _TF.conv_transpose2d(x, w, b, stride=2)


F = _Products()


@contextlib.contextmanager
def bf16_operands(acc64=False):
    global _BF16_OPERANDS, _BF16_ACC64
    prev, _BF16_OPERANDS, _BF16_ACC64 = (_BF16_OPERANDS, _BF16_ACC64), True, acc64
    try:
        yield
    finally:
        _BF16_OPERANDS, _BF16_ACC64 = prev


# ----------------------------------------------------------------------------- config

def make_cfg(**over):
    """Hyper-parameters of conf/defaults/genrl.yaml + conf/env/dmc_pixels.yaml + agent/genrl.yaml."""
    c = dict(
        stoch=32, discrete=32, deter=1024, hidden=1024, units=1024, mlp_layers=4,
        cnn_depth=48, enc_kernels=(4, 4, 4, 4), dec_kernels=(5, 5, 6, 6), img=64, act_dim=10,
        clip_dim=512, n_frames=8,
        kl_free=1.0, kl_forward=False, kl_balance=0.85, kl_scale=0.6,
        conn_kl_free=0.0, conn_kl_forward=True, conn_kl_balance=0.8, conn_loss_scale=1.0,
        lafite_noise=0.5, discount=0.99, lam=0.95, horizon=16, actor_ent=0.0,
        min_std=0.1, max_std=1.0, unimix=0.99, ema_alpha=0.01,
        single_obs_posterior=True, decoder_inputs='stoch', reward_grad=False,     # GenRL defaults

        model_opt=dict(lr=1e-4, eps=1e-8, clip=1000.0, wd=1e-6),
        actor_opt=dict(lr=3e-5, eps=1e-5, clip=100.0, wd=1e-6),
        critic_opt=dict(lr=3e-5, eps=1e-5, clip=100.0, wd=1e-6),
    )
    c.update(over)
    return SimpleNamespace(**c)


# ----------------------------------------------------------------------------- small functions

def symlog(x):  # agent/dreamer_utils.py:13-14
    return torch.sign(x) * torch.log(torch.abs(x) + 1.0)


def symexp(x):  # agent/dreamer_utils.py:16-17
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1.0)


def silu(x):
    return F.silu(x)


def dense_ln_silu(x, w, b, g, beta, eps=1e-5):
    """Linear + LayerNorm(eps 1e-5) + SiLU  (agent/dreamer_utils.py:339,844-859,462-463)."""
    return silu(F.layer_norm(F.linear(x, w, b), (w.shape[0],), g, beta, eps))


def unimix_probs(logits, unimix=0.99):
    """OneHotDist.__init__ (agent/dreamer_utils.py:179-183): softmax, 1 % uniform mix, then the
    renormalisation torch's Categorical(probs=...) applies."""
    p = torch.softmax(logits, -1)
    p = unimix * p + (1 - unimix) * torch.ones_like(p) / p.shape[-1]
    return p / p.sum(-1, keepdim=True)


def probs_to_logits(p):
    eps = torch.finfo(p.dtype).eps
    return torch.log(p.clamp(min=eps, max=1 - eps))


def onehot_sample(logits, q, unimix=0.99):
    """OneHotDist.sample with injected exponential noise `q` (agent/dreamer_utils.py:189-197;
    torch.multinomial on CPU == argmax(p / q)).  Straight-through gradient."""
    p = unimix_probs(logits, unimix)
    idx = torch.argmax(p.detach() / q.reshape(p.shape), -1)
    s = F.one_hot(idx, p.shape[-1]).to(p)
    return s + (p - p.detach())


def onehot_mode(logits, unimix=0.99):
    """OneHotDist.mode (agent/dreamer_utils.py:185-187)."""
    lg = probs_to_logits(unimix_probs(logits, unimix))
    m = F.one_hot(torch.argmax(lg, -1), lg.shape[-1]).to(lg)
    return m.detach() + lg - lg.detach()


def cat_kl(logits_p, logits_q, unimix=0.99):
    """KL(Independent(OneHotDist(p),1) || Independent(OneHotDist(q),1)), summed over the latent
    axis (torch kl_categorical_categorical; agent/dreamer_utils.py:410-420,534-555)."""
    pp, pq = unimix_probs(logits_p, unimix), unimix_probs(logits_q, unimix)
    t = pp * (probs_to_logits(pp) - probs_to_logits(pq))
    return t.sum(-1).sum(-1)


def cat_entropy(logits, unimix=0.99):
    """Independent(OneHotDist,1).entropy()  (agent/dreamer.py:249-250)."""
    p = unimix_probs(logits, unimix)
    return -(probs_to_logits(p) * p).sum(-1).sum(-1)


def kl_loss(post_logit, prior_logit, forward, balance, free):
    """EnsembleRSSM.kl_loss, free_avg=False, balance != 0.5 (agent/dreamer_utils.py:534-555)."""
    lhs, rhs = (prior_logit, post_logit) if forward else (post_logit, prior_logit)
    mix = balance if forward else (1 - balance)
    free_t = torch.tensor([free], dtype=post_logit.dtype)
    value = value_lhs = cat_kl(lhs, rhs.detach())
    value_rhs = cat_kl(lhs.detach(), rhs)
    loss_lhs = torch.maximum(value_lhs, free_t).mean()
    loss_rhs = torch.maximum(value_rhs, free_t).mean()
    return mix * loss_lhs + (1 - mix) * loss_rhs, value


def twohot_buckets(dtype=torch.float32):
    return torch.linspace(-20.0, 20.0, steps=255, dtype=dtype)


def twohot_logprob(logits, x):
    """TwoHotDist.log_prob (agent/dreamer_utils.py:147-171). logits (...,255), x (...,1) -> (...)."""
    buckets = twohot_buckets(logits.dtype)
    x = symlog(x)
    below = torch.sum((buckets <= x[..., None]).to(torch.int32), dim=-1) - 1
    above = len(buckets) - torch.sum((buckets > x[..., None]).to(torch.int32), dim=-1)
    below = torch.clip(below, 0, len(buckets) - 1)
    above = torch.clip(above, 0, len(buckets) - 1)
    equal = below == above
    d_below = torch.where(equal, 1, torch.abs(buckets[below] - x))
    d_above = torch.where(equal, 1, torch.abs(buckets[above] - x))
    total = d_below + d_above
    w_below, w_above = d_above / total, d_below / total
    target = (F.one_hot(below, 255) * w_below[..., None] + F.one_hot(above, 255) * w_above[..., None])
    log_pred = logits - torch.logsumexp(logits, -1, keepdim=True)
    return (target.squeeze(-2) * log_pred).sum(-1)


def twohot_mean(logits):
    """TwoHotDist.mean (agent/dreamer_utils.py:137-140) -> (...,1)."""
    probs = torch.softmax(logits, -1)
    return symexp(torch.sum(probs * twohot_buckets(logits.dtype), dim=-1, keepdim=True))


def lambda_return(reward, value, pcont, bootstrap, lam):
    """lambda_return along axis 0 (agent/dreamer_utils.py:228-253)."""
    next_values = torch.cat([value[1:], bootstrap[None]], 0)
    inputs = reward + pcont * next_values * (1 - lam)
    agg = bootstrap
    outs = []
    for t in reversed(range(reward.shape[0])):
        agg = inputs[t] + pcont[t] * lam * agg
        outs.append(agg)
    return torch.stack(outs[::-1], 0)


def max_cosine_similarity(u, v):  # tools/genrl_utils.py:240-242
    mn = torch.max(torch.norm(u, dim=-1), torch.norm(v, dim=-1)).unsqueeze(-1)
    return torch.sum((u / mn) * (v / mn), dim=-1)


# ----------------------------------------------------------------------------- encoder / decoder

def preprocess_obs(obs_u8):  # agent/dreamer.py:294-295
    return obs_u8 / 255.0 - 0.5


def ch_layer_norm(x, g, b, eps=1e-3):  # ImgChLayerNorm, agent/dreamer_utils.py:1031-1040
    x = x.permute(0, 2, 3, 1)
    x = F.layer_norm(x, (x.shape[-1],), g, b, eps)
    return x.permute(0, 3, 1, 2)


def encoder(p, cfg, obs, prefix='wm.encoder.'):
    """Encoder._cnn (agent/dreamer_utils.py:578-589,618-621). obs (N,3,H,W) float -> (N,E)."""
    x = obs
    for i in range(len(cfg.enc_kernels)):
        x = F.conv2d(x, p[f'{prefix}_conv_model.{3*i}.weight'], p[f'{prefix}_conv_model.{3*i}.bias'], stride=2)
        x = ch_layer_norm(x, p[f'{prefix}_conv_model.{3*i+1}.norm.weight'], p[f'{prefix}_conv_model.{3*i+1}.norm.bias'])
        x = silu(x)
    return x.reshape(x.shape[0], -1)


def decoder(p, cfg, feat, prefix='wm.heads.decoder.'):
    """Decoder._cnn (agent/dreamer_utils.py:654-671,695-706). feat (N,S) -> mean image (N,3,H,W)."""
    x = F.linear(feat, p[f'{prefix}_conv_in.0.weight'], p[f'{prefix}_conv_in.0.bias'])
    x = x.reshape(-1, 32 * cfg.cnn_depth, 1, 1)
    n = len(cfg.dec_kernels)
    for i in range(n):
        x = F.conv_transpose2d(x, p[f'{prefix}_conv_model.{3*i}.weight'], p[f'{prefix}_conv_model.{3*i}.bias'], stride=2)
        if i != n - 1:
            x = ch_layer_norm(x, p[f'{prefix}_conv_model.{3*i+1}.norm.weight'], p[f'{prefix}_conv_model.{3*i+1}.norm.bias'])
            x = silu(x)
    return x


def mlp_trunk(p, prefix, x, layers):
    """MLP body (agent/dreamer_utils.py:739-747): layers x [Linear+LN+SiLU]."""
    for i in range(layers):
        x = dense_ln_silu(x, p[f'{prefix}dense{i}.weight'], p[f'{prefix}dense{i}.bias'],
                          p[f'{prefix}norm{i}._layer.weight'], p[f'{prefix}norm{i}._layer.bias'])
    return x


def mlp_head(p, prefix, x, layers):
    """MLP with a twohot head -> 255 logits."""
    x = mlp_trunk(p, prefix, x, layers)
    return F.linear(x, p[f'{prefix}_out._out.weight'], p[f'{prefix}_out._out.bias'])


def actor_stats(p, cfg, x, prefix):
    """MLP + DistLayer('normal') (agent/dreamer_utils.py:802-819): mean=tanh, std in [min,max]."""
    x = mlp_trunk(p, prefix, x, cfg.mlp_layers)
    out = F.linear(x, p[f'{prefix}_out._out.weight'], p[f'{prefix}_out._out.bias'])
    std = F.linear(x, p[f'{prefix}_out._std.weight'], p[f'{prefix}_out._std.bias'])
    mean = torch.tanh(out)
    std = (cfg.max_std - cfg.min_std) * torch.sigmoid(std + 2.0) + cfg.min_std
    return mean, std


def normal_entropy(std):  # Independent(Normal,1).entropy()
    return (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(-1)


# ----------------------------------------------------------------------------- RSSM

def gru_cell(p, prefix, x, h):
    """GRUCell.forward with norm=True, update_bias=-1 (agent/dreamer_utils.py:771-785)."""
    parts = F.linear(torch.cat([x, h], -1), p[f'{prefix}_cell._layer.weight'])
    parts = F.layer_norm(parts, (parts.shape[-1],), p[f'{prefix}_cell._norm.weight'], p[f'{prefix}_cell._norm.bias'], 1e-5)
    reset, cand, update = torch.chunk(parts, 3, -1)
    reset = torch.sigmoid(reset)
    cand = torch.tanh(reset * cand)
    update = torch.sigmoid(update - 1.0)
    return update * cand + (1 - update) * h


def prior_logits(p, cfg, prefix, deter):
    """get_stoch_stats_from_deter_state, ensemble=1 (agent/dreamer_utils.py:475-521)."""
    x = dense_ln_silu(deter, p[f'{prefix}_ensemble_img_out.0.0.weight'], p[f'{prefix}_ensemble_img_out.0.0.bias'],
                      p[f'{prefix}_ensemble_img_out.0.1._layer.weight'], p[f'{prefix}_ensemble_img_out.0.1._layer.bias'])
    lg = F.linear(x, p[f'{prefix}_ensemble_img_dist.0.weight'], p[f'{prefix}_ensemble_img_dist.0.bias'])
    return lg.reshape(list(lg.shape[:-1]) + [cfg.stoch, cfg.discrete])


def img_step(p, cfg, prefix, stoch, deter, action, q=None):
    """EnsembleRSSM.img_step (agent/dreamer_utils.py:459-473). stoch (R,S,K); q exponential noise
    (R*S,K) or None -> mode."""
    x = torch.cat([stoch.reshape(stoch.shape[0], -1), action], -1)
    x = dense_ln_silu(x, p[f'{prefix}_img_in.0.weight'], p[f'{prefix}_img_in.0.bias'],
                      p[f'{prefix}_img_in.1._layer.weight'], p[f'{prefix}_img_in.1._layer.bias'])
    deter = gru_cell(p, prefix, x, deter)
    logit = prior_logits(p, cfg, prefix, deter)
    new_stoch = onehot_sample(logit, q, cfg.unimix) if q is not None else onehot_mode(logit, cfg.unimix)
    return dict(stoch=new_stoch, deter=deter, logit=logit)


def post_logits(p, cfg, embed, prefix='wm.rssm.', deter=None):
    """get_post_stoch (agent/dreamer_utils.py:442-457); without single_obs_posterior the input is
    cat([deter, embed])."""
    if not cfg.single_obs_posterior:
        embed = torch.cat([deter, embed], -1)
    x = dense_ln_silu(embed, p[f'{prefix}_obs_out.0.weight'], p[f'{prefix}_obs_out.0.bias'],
                      p[f'{prefix}_obs_out.1._layer.weight'], p[f'{prefix}_obs_out.1._layer.bias'])
    lg = F.linear(x, p[f'{prefix}_obs_dist.weight'], p[f'{prefix}_obs_dist.bias'])
    return lg.reshape(list(lg.shape[:-1]) + [cfg.stoch, cfg.discrete])


def observe(p, cfg, embed, action, is_first, noise, prefix='wm.rssm.'):
    """EnsembleRSSM.observe / obs_step (agent/dreamer_utils.py:362-371,432-440).
    embed (B,T,E), action (B,T,A), is_first (B,T) bool.
    noise: 'prior_q' and 'post_q', each (T, B*S, K) exponential draws (order of the reference's
    RNG consumption: prior sample then posterior sample, per step)."""
    B, T = action.shape[:2]
    stoch = torch.zeros(B, cfg.stoch, cfg.discrete)
    deter = torch.zeros(B, cfg.deter)
    posts, priors = [], []
    for t in range(T):
        a = action[:, t]
        if is_first[:, t].any():
            m = 1.0 - is_first[:, t].float()
            stoch = torch.einsum('b,b...->b...', m, stoch)
            deter = torch.einsum('b,b...->b...', m, deter)
            a = torch.einsum('b,b...->b...', m, a)
        prior = img_step(p, cfg, prefix, stoch, deter, a, noise['prior_q'][t])
        plog = post_logits(p, cfg, embed[:, t], prefix, prior['deter'])
        pst = onehot_sample(plog, noise['post_q'][t], cfg.unimix)
        post = dict(stoch=pst, deter=prior['deter'], logit=plog)
        posts.append(post); priors.append(prior)
        stoch, deter = post['stoch'], post['deter']
    stack = lambda L: {k: torch.stack([d[k] for d in L], 1) for k in L[0]}
    return stack(posts), stack(priors)


def get_feat(state):  # agent/dreamer_utils.py:405-408
    s = state['stoch']
    return torch.cat([s.reshape(list(s.shape[:-2]) + [-1]), state['deter']], -1)


# ----------------------------------------------------------------------------- world-model loss

def wm_loss(p, cfg, batch, noise):
    """WorldModel.loss (agent/dreamer.py:219-252) for grad_heads=[decoder], decoder_inputs=stoch.
    batch: observation u8 (B,T,3,H,W), action, reward (B,T,1), is_first (B,T)."""
    B, T = batch['action'].shape[:2]
    obs = preprocess_obs(batch['observation'])
    embed = encoder(p, cfg, obs.reshape((-1,) + tuple(obs.shape[2:]))).reshape(B, T, -1)
    post, prior = observe(p, cfg, embed, batch['action'], batch['is_first'], noise)
    kl, kl_value = kl_loss(post['logit'], prior['logit'], cfg.kl_forward, cfg.kl_balance, cfg.kl_free)
    feat = get_feat(post)
    dec_in = post['stoch'].reshape(B * T, -1) if cfg.decoder_inputs == 'stoch' else feat.reshape(B * T, -1)
    recon = decoder(p, cfg, dec_in).reshape(obs.shape)
    like_obs = -((recon - obs) ** 2).sum([2, 3, 4])               # MSEDist.log_prob, agg sum
    rew_logits = mlp_head(p, 'wm.heads.reward.', feat if cfg.reward_grad else feat.detach(), cfg.mlp_layers)
    like_rew = twohot_logprob(rew_logits, batch['reward'])
    losses = dict(kl=kl, observation=-like_obs.mean(), reward=-like_rew.mean())
    model_loss = cfg.kl_scale * losses['kl'] + losses['observation'] + losses['reward']
    outs = dict(embed=embed, feat=feat, post=post, prior=prior,
                likes=dict(observation=like_obs, reward=like_rew), kl=kl_value, recon=recon)
    metrics = {f'{k}_loss': v for k, v in losses.items()}
    metrics['model_kl'] = kl_value.mean()
    metrics['prior_ent'] = cat_entropy(prior['logit'], cfg.unimix).mean()
    metrics['post_ent'] = cat_entropy(post['logit'], cfg.unimix).mean()
    return model_loss, outs, metrics


# ----------------------------------------------------------------------------- connector

def residual_linear(p, prefix, x, norm=True, act=True):
    """ResidualLinear, prenorm=False (agent/video_utils.py:8-25)."""
    h = F.linear(x, p[f'{prefix}layer.weight'], p[f'{prefix}layer.bias'])
    if norm:
        h = F.layer_norm(h, (h.shape[-1],), p[f'{prefix}norm_layer._layer.weight'], p[f'{prefix}norm_layer._layer.bias'], 1e-5)
    if act:
        h = silu(h)
    if f'{prefix}res_proj.weight' in p:
        r = F.linear(x, p[f'{prefix}res_proj.weight'], p[f'{prefix}res_proj.bias'])
    else:
        r = x
    return h + r


def aligner(p, x, prefix='wm.connector.aligner.'):
    """UNetDenoiser, n_layers=2 (agent/video_utils.py:27-61)."""
    d0 = residual_linear(p, f'{prefix}down.0.', x)
    d1 = residual_linear(p, f'{prefix}down.1.', d0)
    m = residual_linear(p, f'{prefix}mid.0.', d1)
    m = residual_linear(p, f'{prefix}mid.1.', m)
    u = residual_linear(p, f'{prefix}up.0.', torch.cat([m, d1], -1), norm=False, act=False)
    u = residual_linear(p, f'{prefix}up.1.', torch.cat([u, d0], -1))
    return u


def connector_action(cfg, video_embed):
    """VideoSSM.get_action with rescale_embeds, no temporal embeds (agent/video_utils.py:114-125)."""
    z = torch.zeros(list(video_embed.shape[:-1]) + [cfg.n_frames])
    return torch.cat([video_embed * math.sqrt(cfg.clip_dim), z], -1)


def connector_initial(p, cfg, action0, q, prefix='wm.connector.'):
    """VideoSSM.initial with learn_initial (agent/video_utils.py:100-112)."""
    x = dense_ln_silu(action0, p[f'{prefix}initial_state_pred.0.weight'], p[f'{prefix}initial_state_pred.0.bias'],
                      p[f'{prefix}initial_state_pred.1._layer.weight'], p[f'{prefix}initial_state_pred.1._layer.bias'])
    x = dense_ln_silu(x, p[f'{prefix}initial_state_pred.3.weight'], p[f'{prefix}initial_state_pred.3.bias'],
                      p[f'{prefix}initial_state_pred.4._layer.weight'], p[f'{prefix}initial_state_pred.4._layer.bias'])
    deter = F.linear(x, p[f'{prefix}initial_state_pred.6.weight'], p[f'{prefix}initial_state_pred.6.bias'])
    logit = prior_logits(p, cfg, prefix, deter)
    stoch = onehot_sample(logit, q, cfg.unimix)
    return dict(stoch=stoch, deter=deter, logit=logit)


def connector_loss(p, cfg, clip_video, wm_post, noise, prefix='wm.connector.'):
    """VideoSSM.update (agent/video_utils.py:127-207).
    noise: 'clip_eps' (B,T,512) normal; 'init_q' (B*S,K); 'step_q' (T,B*S,K);
    'ikl_init_q', 'ikl_step_q' (B*(T/8-1)*S, K)."""
    nf = cfg.n_frames
    B, T = clip_video.shape[:2]
    ve = clip_video[:, nf - 1::nf]
    ve = ve.reshape(B, T // nf, 1, -1).repeat(1, 1, nf, 1).reshape(B, T, -1)
    orig = ve
    nn_ = F.normalize(noise['clip_eps'], dim=-1)
    ve = (1 - cfg.lafite_noise) * ve + cfg.lafite_noise * nn_
    ve = F.normalize(ve, dim=-1)
    den = F.normalize(aligner(p, ve, f'{prefix}aligner.'), dim=-1)
    denoising_loss = 1 - F.cosine_similarity(den, orig, dim=-1).mean()
    ve = orig
    acts = connector_action(cfg, ve)
    post = {k: v.detach() for k, v in wm_post.items()}
    priors = []
    for t in range(T):
        a = acts[:, t]
        if t == 0:
            prev = connector_initial(p, cfg, a, noise['init_q'], prefix)
        else:
            prev = dict(prior)
            prev['stoch'] = post['stoch'][:, t - 1]
        prior = img_step(p, cfg, prefix, prev['stoch'], prev['deter'], a, noise['step_q'][t])
        priors.append(prior)
    prior_logit = torch.stack([d['logit'] for d in priors], 1)
    kl, kl_value = kl_loss(post['logit'], prior_logit, cfg.conn_kl_forward, cfg.conn_kl_balance, cfg.conn_kl_free)
    loss = denoising_loss + cfg.conn_loss_scale * kl
    metrics = dict(aligner_cosine_distance=denoising_loss, connector_kl=kl_value.mean())
    # initial KL (metric only; consumes noise)
    G = T // nf
    ve2 = ve.reshape(B, G, nf, -1)[:, 1:, 0].reshape(B * (G - 1), -1)
    a2 = connector_action(cfg, ve2)
    post2 = {k: v.reshape(B, G, nf, *v.shape[2:])[:, 1:, 0].reshape(B * (G - 1), *v.shape[2:]) for k, v in post.items()}
    prev = connector_initial(p, cfg, a2, noise['ikl_init_q'], prefix)
    pr = img_step(p, cfg, prefix, prev['stoch'], prev['deter'], a2, noise['ikl_step_q'])
    _, ikl = kl_loss(post2['logit'], pr['logit'], cfg.conn_kl_forward, cfg.conn_kl_balance, cfg.conn_kl_free)
    metrics['connector_initial_kl'] = ikl.mean()
    return loss, metrics


def video_imagine_target(p, cfg, text_feat, n_rows, steps, init_q, prefix='wm.connector.'):
    """The `unconditional_target` of video_text_reward with skip_first_target, sample_for_target
    False (tools/genrl_utils.py:305-309) = VideoSSM.video_imagine(sample=False, denoise=True,
    reset_every_n_frames=False) (agent/video_utils.py:209-240). Returns time-major dict (steps,N,..)."""
    ve = text_feat.reshape(1, 1, -1).repeat(n_rows, steps + 1, 1)
    ve = F.normalize(aligner(p, ve, f'{prefix}aligner.'), dim=-1)
    acts = connector_action(cfg, ve)
    st = connector_initial(p, cfg, acts[:, 0], init_q, prefix)
    outs = []
    for t in range(steps + 1):
        st = img_step(p, cfg, prefix, st['stoch'], st['deter'], acts[:, t], None)
        outs.append(st)
    return {k: torch.stack([d[k] for d in outs[1:]], 0) for k in outs[0]}


# ----------------------------------------------------------------------------- imagination + AC

def imagine(p, cfg, start, noise, actor_prefix='_imag_behavior.actor.', rssm_prefix='wm.rssm.'):
    """WorldModel.imagine (agent/dreamer.py:254-287), no discount head -> weight == 1.
    start: post dict (B,T,..) (detached). noise: 'act_eps' (H,N,A) normal, 'step_q' (H,N*S,K).
    (The reference's throw-away first policy sample (agent/dreamer.py:259-260) only shapes a zero
    action; it consumes RNG but not values.)"""
    flat = lambda x: x.reshape([-1] + list(x.shape[2:]))
    st = {k: flat(v) for k, v in start.items()}
    N = st['deter'].shape[0]
    seq = dict(stoch=[st['stoch']], deter=[st['deter']], logit=[st['logit']],
               feat=[get_feat(st)], action=[torch.zeros(N, cfg.act_dim)])
    for h in range(cfg.horizon):
        mean, std = actor_stats(p, cfg, seq['feat'][-1].detach(), actor_prefix)
        action = mean + std * noise['act_eps'][h]
        s = img_step(p, cfg, rssm_prefix, seq['stoch'][-1], seq['deter'][-1], action, noise['step_q'][h])
        for k, v in {**s, 'action': action, 'feat': get_feat(s)}.items():
            seq[k].append(v)
    seq = {k: torch.stack(v, 0) for k, v in seq.items()}
    disc = torch.ones(list(seq['feat'].shape[:-1]) + [1])
    seq['discount'] = disc * cfg.discount
    seq['weight'] = torch.cumprod(torch.cat([torch.ones_like(disc[:1]), disc[:-1]], 0), 0)
    return seq


def conv_in(p, stoch):  # decoder._conv_in[0] on flattened stoch (tools/genrl_utils.py:253-256)
    s = stoch.reshape(list(stoch.shape[:-2]) + [-1])
    return F.linear(s, p['wm.heads.decoder._conv_in.0.weight'], p['wm.heads.decoder._conv_in.0.bias'])


def video_text_reward(p, cfg, seq_stoch, target_stoch):
    """video_text_reward with score_fn=max_cosine, align_sequence (tools/genrl_utils.py:344-366).
    seq_stoch (T,N,S,K) agent; target_stoch (T,N,S,K). Returns (T,N,1)."""
    T = seq_stoch.shape[0]
    nf = cfg.n_frames
    ct_short = conv_in(p, target_stoch[:nf])
    scores = []
    for t in range(T - nf):
        ca = conv_in(p, seq_stoch[t:t + nf])
        scores.append(max_cosine_similarity(ct_short, ca).mean(0))
    align = torch.stack(scores, 0)
    best = F.one_hot(torch.argmax(align, 0), T)
    ts_idx = torch.clip(torch.cumsum(torch.cumsum(best, 1), 1) - 1, min=0).T      # (T,N)
    idx = ts_idx[..., None, None].repeat(1, 1, target_stoch.shape[-2], target_stoch.shape[-1])
    new_t = torch.gather(target_stoch, 0, idx)
    return max_cosine_similarity(conv_in(p, new_t), conv_in(p, seq_stoch)).unsqueeze(-1), ts_idx


def quantile_ema(target, ema_vals, alpha):
    """RewardEMA.__call__ (agent/dreamer_utils.py:1022-1029). Returns (offset, scale, new_ema)."""
    q = torch.quantile(target.detach().flatten(), torch.tensor([0.05, 0.95]))
    new = alpha * q + (1 - alpha) * ema_vals
    scale = torch.clip(new[1] - new[0], min=1.0)
    return new[0], scale, new


def actor_critic_losses(p, cfg, seq, reward, ema_vals, prefix='_imag_behavior.'):
    """ActorCritic.target/actor_loss/critic_loss (agent/dreamer.py:392-453), actor_grad=dynamics,
    reward_ema, slow target.  Returns actor_loss, critic_loss_fn inputs, metrics, new ema."""
    value = twohot_mean(mlp_head(p, f'{prefix}_target_critic.', seq['feat'], cfg.mlp_layers))
    target = lambda_return(reward[:-1], value[:-1], seq['discount'][:-1], value[-1], cfg.lam)
    mets = dict(critic_slow=value.mean(), critic_target=target.mean())
    mean, std = actor_stats(p, cfg, seq['feat'][:-2].detach(), f'{prefix}actor.')
    offset, scale, new_ema = quantile_ema(target, ema_vals, cfg.ema_alpha)
    normed = (target - offset.detach()) / scale.detach()
    mets['normed_target_mean'] = normed.mean(); mets['normed_target_std'] = normed.std()
    mets['reward_ema_005'] = new_ema[0]; mets['reward_ema_095'] = new_ema[1]
    ent = normal_entropy(std)[:, :, None]
    objective = normed[1:] + cfg.actor_ent * ent
    mets['actor_ent'] = ent.mean(); mets['actor_ent_scale'] = cfg.actor_ent
    actor_loss = -(seq['weight'].detach()[:-2] * objective).mean()
    # critic
    feat = seq['feat'][:-1].detach()
    logits = mlp_head(p, f'{prefix}critic.', feat, cfg.mlp_layers)
    critic_loss = -(twohot_logprob(logits, target.detach())[:, :, None] * seq['weight'].detach()[:-1]).mean()
    mets['critic'] = twohot_mean(logits).mean()
    return actor_loss, critic_loss, target, mets, new_ema


def stream_norm_metrics(reward):  # StreamNorm.__call__ with momentum 1 (agent/dreamer_utils.py:956-964)
    return dict(reward_mean=reward.mean(), reward_std=reward.std(),
                reward_normed_mean=reward.mean(), reward_normed_std=reward.std())


# ----------------------------------------------------------------------------- optimiser

def global_grad_norm(grads):
    return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0)


def optimizer_step(params, grads, state, lr, eps, clip, wd, decay_only=()):
    """Optimizer.__call__ after backward (agent/dreamer_utils.py:910-923): clip by global norm,
    multiplicative weight decay on every handed parameter, Adam(0.9, 0.999).
    params/grads: dict name->tensor; state: dict name->(step, m, v), updated in place.
    decay_only: names that receive weight decay but have no gradient.
    Returns the pre-clip gradient norm."""
    names = list(grads.keys())
    norm = global_grad_norm([grads[n] for n in names])
    coef = torch.clamp(clip / (norm + 1e-6), max=1.0)
    for n in list(names) + list(decay_only):
        params[n] = (1 - wd) * params[n]
    for n in names:
        g = grads[n] * coef
        step, m, v = state.get(n, (0, torch.zeros_like(g), torch.zeros_like(g)))
        step += 1
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
        denom = (v.sqrt() / math.sqrt(bc2)) + eps
        params[n] = params[n] - (lr / bc1) * (m / denom)
        state[n] = (step, m, v)
    return norm


def env_reward(p, cfg, seq):
    """env_reward (agent/dreamer.py:16-17): the reward head's mean on imagined features."""
    return twohot_mean(mlp_head(p, 'wm.heads.reward.', seq['feat'], cfg.mlp_layers))
