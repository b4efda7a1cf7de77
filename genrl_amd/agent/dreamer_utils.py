"""MI355X-native building blocks behind the reference's `agent/dreamer_utils.py` API.

Class names, constructor signatures, sub-module and parameter names follow the reference
(mazpie/genrl agent/dreamer_utils.py; weight contract in SURVEY.md §8a) so that `state_dict()`s and
pickled agents are interchangeable; every forward/backward runs in the hand-written HIP kernels
of libgenrl_hip.so (genrl_amd/ops.py).  nn.Module containers (nn.Linear, nn.Conv2d, nn.LayerNorm)
are used only as *parameter holders*: their torch forward is never called on the hot path.

Only what the GenRL path configures is implemented (norm 'layer'/'none', act SiLU, discrete
latents, GRU cell, dists mse / twohot / normal / onehot); other reference options raise.
"""
import contextlib
import re

import numpy as np
import torch
import torch.nn as nn

import os
from .. import noise, ops, ops_conv_planes, ops_planes, streams
from .. import planes as pl          # (module; `planes=` below are operand handles)


class Module(nn.Module):
    """nn.Module whose load_state_dict also drops the cached weight planes (genrl_amd/planes.py)"""
    def load_state_dict(self, *args, **kwargs):
        pl.invalidate()
        return super().load_state_dict(*args, **kwargs)


def symlog(x):  # ref :13-14
    return torch.sign(x) * torch.log(torch.abs(x) + 1.0)


def symexp(x):  # ref :16-17
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1.0)


# ----------------------------------------------------------------------------- distributions

class MSEDist:
    """ref :62-83.  `log_prob` accepts the raw uint8 frames (preprocess fused into the kernel)."""
    def __init__(self, mode, agg='sum'):
        assert agg == 'sum'
        self._mode = mode

    @property
    def mean(self):
        return self._mode

    def mode(self):
        return self._mode

    def log_prob(self, value):
        assert self._mode.shape == value.shape, (self._mode.shape, value.shape)
        if value.dtype != torch.uint8:           # already-preprocessed floats: undo (x+0.5)*255 exactly
            value = torch.round((value + 0.5) * 255.0).to(torch.uint8)
        lead = self._mode.shape[:-3]
        like = ops.mse_like(self._mode.reshape((-1,) + tuple(self._mode.shape[-3:])),
                            value.reshape((-1,) + tuple(value.shape[-3:])))
        return like.reshape(lead)


class TwoHotDist:
    """ref :120-171 (255 symlog buckets in [-20, 20])."""
    def __init__(self, logits, low=-20.0, high=20.0):
        assert logits.shape[-1] == 255
        assert low == -20.0 and high == 20.0
        self.logits = logits

    @property
    def mean(self):
        return ops.twohot_mean(self.logits)

    @property
    def mode(self):
        return self.mean

    def log_prob(self, x):
        return ops.twohot_logprob(self.logits, x)


class OneHotDist:
    """ref :177-197 wrapped in Independent(.,1) (ref :413-415): unimix categorical latents."""
    def __init__(self, logits, site='onehot'):
        self.logits_raw = logits
        self.site = site

    def sample(self, sample_shape=()):
        lg = self.logits_raw
        K = lg.shape[-1]
        q = noise.draw('exp', self.site, (lg.numel() // K, K), lg.device)
        return ops.onehot_sample(lg, q)

    def mode(self):
        return ops.onehot_mode(self.logits_raw)

    @property
    def probs(self):
        """Class probabilities after the 1 % uniform mix (what OneHotCategorical(probs=...) holds, ref :179-183)."""
        K = self.logits_raw.shape[-1]
        return ops.UNIMIX * torch.softmax(self.logits_raw.float(), -1) + (1.0 - ops.UNIMIX) / K

    @property
    def mean(self):          # OneHotCategorical.mean; train.py:297 stores it as the start 'logit' of data-free rollouts
        return self.probs

    def entropy(self):
        return ops.cat_entropy(self.logits_raw)


def kl_divergence(p, q):
    return ops.cat_kl(p.logits_raw, q.logits_raw)


class NormalDist:
    """Independent(Normal(tanh(out), std), 1) of DistLayer 'normal' (ref :814-819)."""
    def __init__(self, raw, min_std, max_std, site='actor'):
        self.raw, self.min_std, self.max_std, self.site = raw, min_std, max_std, site

    def sample(self):
        A = self.raw.shape[-1] // 2
        eps = noise.draw('normal', self.site, tuple(self.raw.shape[:-1]) + (A,), self.raw.device)
        return ops.actor_sample(self.raw, eps, self.min_std, self.max_std)

    rsample = sample

    @property
    def mean(self):
        return ops.actor_mean_std(self.raw, self.min_std, self.max_std)[0]

    def entropy(self):
        std = ops.actor_mean_std(self.raw, self.min_std, self.max_std)[1]
        return (0.5 + 0.5 * np.log(2 * np.pi) + torch.log(std)).sum(-1)


# ----------------------------------------------------------------------------- scans (API parity)

def lambda_return(reward, value, pcont, bootstrap, lambda_, axis):
    """ref :228-253 (axis 0, constant pcont).  reward/value (H,N,1); bootstrap (N,1)."""
    assert axis == 0
    disc = float(pcont) if isinstance(pcont, (int, float)) else float(pcont.flatten()[0])
    v = torch.cat([value, bootstrap[None]], 0)
    return ops.lambda_return(reward, v, disc, lambda_)


def static_scan(fn, inputs, start, reverse=False, unpack=False):
    """ref :255-300 (generic python scan; the hot loops do not use it)."""
    last, outs = start, None
    for i in range(inputs[0].shape[0]):
        last = fn(last, *[x[i] for x in inputs]) if unpack else fn(last, tuple(x[i] for x in inputs))
        items = [last] if isinstance(last, dict) else list(last)
        if outs is None:
            outs = [{k: [v] for k, v in it.items()} if isinstance(it, dict) else [it] for it in items]
        else:
            for o, it in zip(outs, items):
                if isinstance(it, dict):
                    for k, v in it.items():
                        o[k].append(v)
                else:
                    o.append(it)
    return [{k: torch.stack(v, 0) for k, v in o.items()} if isinstance(o, dict) else torch.stack(o, 0) for o in outs]


# ----------------------------------------------------------------------------- layers

def get_act(name):
    if name == 'none':
        return nn.Identity()
    if hasattr(nn, name):
        return getattr(nn, name)()
    raise NotImplementedError(name)


_zero_cache = {}
_one_cache = {}


def _one_like(t):
    key = (tuple(t.shape), str(t.device), t.dtype)
    if key not in _one_cache:
        _one_cache[key] = torch.ones(tuple(t.shape), device=t.device, dtype=t.dtype)
    return _one_cache[key]



def _zeros(shape, dev):
    """A zero tensor that is never written (initial RSSM states are replaced, not updated in place): cached per shape so
    that a training step does not spend a fill launch per state entry."""
    key = (tuple(shape), str(dev))
    if key not in _zero_cache:
        _zero_cache[key] = torch.zeros(tuple(shape), device=dev)
    return _zero_cache[key]


class NormLayer(Module):  # ref :844-859
    def __init__(self, name, dim=None):
        super().__init__()
        if name == 'none':
            self._layer = None
        elif name == 'layer':
            assert dim is not None
            self._layer = nn.LayerNorm(dim)
        else:
            raise NotImplementedError(name)

    def forward(self, features):
        if self._layer is None:
            return features
        return ops.ln_act(features, self._layer.weight, self._layer.bias, self._layer.eps, act=False)


class ImgChLayerNorm(nn.Module):  # ref :1031-1040 (parameter holder; applied on NHWC rows)
    def __init__(self, ch, eps=1e-03):
        super().__init__()
        self.norm = torch.nn.LayerNorm(ch, eps=eps)

    def forward(self, x):  # x NCHW, API parity only (hot path works on NHWC)
        y = ops.ln_act(x.permute(0, 2, 3, 1).contiguous(), self.norm.weight, self.norm.bias, self.norm.eps, act=False)
        return y.permute(0, 3, 1, 2)


def _dense_ln_silu(x, lin, norm, x2=None, planes=None):
    """Linear (+ second concatenated input) + LayerNorm + SiLU with the reference's layer objects.  planes: h2 planes of
    the inputs (genrl_amd/planes.py) when the caller has them."""
    if norm._layer is None:
        return ops_silu(None)
    rows = x.numel() // x.shape[-1]
    if (pl.ENABLED and x.is_cuda and rows >= ops_planes.min_rows() and lin.weight.shape[0] % 4 == 0
            and os.environ.get('GENRL_PLANES_MLP', '1') != '0'):  # (-0.75 ms/step at c2 with the BK-64 128x128 tile)
        return ops_planes.dense_ln_act(x, x2, lin.weight, lin.bias, norm._layer.weight, norm._layer.bias, norm._layer.eps,
                                   planes=planes)
    return ops.dense_ln_act(x, x2, lin.weight, lin.bias, norm._layer.weight, norm._layer.bias, norm._layer.eps)


def ops_silu(x):
    raise NotImplementedError('norm: none is not on the GenRL path (conf/defaults/genrl.yaml uses layer)')


class GRUCell(Module):  # ref :750-785
    def __init__(self, inp_size, size, norm=False, act='Tanh', update_bias=-1, device='cuda', **kwargs):
        super().__init__()
        assert norm and act == 'Tanh' and update_bias == -1, 'only the GenRL configuration is implemented'
        self._inp_size, self._size = inp_size, size
        self._update_bias = update_bias
        self.device = device
        self._layer = nn.Linear(inp_size + size, 3 * size, bias=False, **kwargs)
        self._norm = nn.LayerNorm(3 * size)

    def get_initial_state(self, inputs=None, batch_size=None, dtype=None):
        return _zeros((batch_size, self._size), self.device)

    @property
    def state_size(self):
        return self._size

    def forward(self, inputs, deter_state):
        out = ops.gru_step(inputs, deter_state[0], self._layer.weight, self._norm.weight, self._norm.bias)
        return out, [out]


class DistLayer(Module):  # ref :787-841
    def __init__(self, in_dim, shape, dist='mse', min_std=0.1, max_std=1.0, init_std=0.0, bias=True):
        super().__init__()
        self._in_dim = in_dim
        self._shape = shape if type(shape) in [list, tuple] else [shape]
        self._dist, self._min_std, self._init_std, self._max_std = dist, min_std, init_std, max_std
        self._out = nn.Linear(in_dim, int(np.prod(shape)), bias=bias)
        if dist == 'normal':
            self._std = nn.Linear(in_dim, int(np.prod(shape)))
        elif dist not in ('twohot', 'mse', 'onehot'):
            raise NotImplementedError(dist)

    def raw(self, inputs):
        h = getattr(inputs, '_planes', None)          # the trunk's last layer left its output's operand planes (ops_planes)
        rows = inputs.numel() // inputs.shape[-1]
        if (h is not None and pl.ENABLED and self._dist != 'normal' and rows >= ops_planes.min_rows()
                and os.environ.get('GENRL_PLANES_LINEAR', '1') != '0'):
            return ops_planes.linear(inputs, self._out.weight, self._out.bias, h)
        out = ops.linear(inputs, self._out.weight, self._out.bias)
        if self._dist == 'normal':
            std = ops.linear(inputs, self._std.weight, self._std.bias)
            return torch.cat([out, std], -1)
        return out

    def forward(self, inputs):
        raw = self.raw(inputs)
        if self._dist == 'normal':
            return NormalDist(raw, self._min_std, self._max_std)
        if self._dist == 'twohot':
            return TwoHotDist(raw)
        if self._dist == 'mse':
            return MSEDist(raw.reshape(list(inputs.shape[:-1]) + list(self._shape)))
        if self._dist == 'onehot':
            return OneHotDist(raw)
        raise NotImplementedError(self._dist)


class MLP(Module):  # ref :718-747
    def __init__(self, in_shape, shape, layers, units, act='SiLU', norm='none', **out):
        super().__init__()
        assert act == 'SiLU'
        self._in_shape = in_shape
        if out['dist'] == 'twohot':
            shape = 255
        self._shape = (shape,) if isinstance(shape, int) else shape
        self._layers, self._units, self._norm = layers, units, norm
        last_units = in_shape
        for index in range(self._layers):
            self.add_module(f'dense{index}', nn.Linear(last_units, units, bias=norm != 'none'))
            self.add_module(f'norm{index}', NormLayer(norm, units))
            last_units = units
        self._out = DistLayer(units, shape, **out)

    def trunk(self, features, features2=None, planes=None):
        """features2: optional second input concatenated after `features` (feat = [stoch, deter])
        consumed without materialising the concatenation.  planes: h2 planes of the inputs, if the caller has them."""
        x = features.reshape([-1, features.shape[-1]])
        x2 = features2.reshape([-1, features2.shape[-1]]) if features2 is not None else None
        for index in range(self._layers):
            x = _dense_ln_silu(x, getattr(self, f'dense{index}'), getattr(self, f'norm{index}'), x2, planes=planes)
            x2 = planes = None
        out = x.reshape(list(features.shape[:-1]) + [x.shape[-1]])
        h = getattr(x, '_planes', None)        # (reshape returns a new tensor object: carry the operand planes of the rows along)
        if h is not None:
            out._planes = h
        return out

    def forward(self, features, features2=None, planes=None):
        return self._out(self.trunk(features, features2, planes))


# ----------------------------------------------------------------------------- encoder / decoder

def _conv_sizes(size, kernels, transposed=False):
    out = []
    for k in kernels:
        size = 2 * (size - 1) + k if transposed else (size - k) // 2 + 1
        out.append(size)
    return out


class Encoder(Module):  # ref :558-628
    def __init__(self, shapes, cnn_keys=r'.*', mlp_keys=r'.*', act='SiLU', norm='none',
                 cnn_depth=48, cnn_kernels=(4, 4, 4, 4), mlp_layers=[400, 400, 400, 400], symlog_inputs=False):
        super().__init__()
        self.shapes = shapes
        self.cnn_keys = [k for k, v in shapes.items() if re.match(cnn_keys, k) and len(v) == 3]
        self.mlp_keys = [k for k, v in shapes.items() if re.match(mlp_keys, k) and len(v) == 1]
        assert act == 'SiLU' and norm == 'layer', 'GenRL path: SiLU + layer norm'
        assert len(self.mlp_keys) == 0, 'proprio MLP inputs are not on the GenRL pixel path'
        self._cnn_depth, self._cnn_kernels = cnn_depth, cnn_kernels
        layers = []
        for i, kernel in enumerate(self._cnn_kernels):
            prev_depth = 3 if i == 0 else 2 ** (i - 1) * self._cnn_depth
            depth = 2 ** i * self._cnn_depth
            layers += [nn.Conv2d(prev_depth, depth, kernel, stride=2), ImgChLayerNorm(depth), get_act(act)]
        self._conv_model = nn.Sequential(*layers)

    def forward(self, data):
        key = self.cnn_keys[0]
        x = data[key]
        shape = self.shapes[key]
        batch_dims = x.shape[:-len(shape)]
        x = x.reshape((-1,) + tuple(x.shape)[len(batch_dims):])
        out = self._cnn(x)
        return out.reshape(tuple(batch_dims) + tuple(out.shape[1:]))

    def _cnn(self, x):
        """x: uint8 NCHW frames (x/255-0.5 fused into the first layer's patch gather) or float NCHW."""
        if x.dtype != torch.uint8:
            x = x.permute(0, 2, 3, 1).contiguous()
        conv2d = ops_conv_planes.conv2d_s2 if ops_conv_planes.active(x) else ops.conv2d_s2     # (products on h2 planes where they fit)
        last = len(self._cnn_kernels) - 1
        for i in range(last + 1):
            conv, ln = self._conv_model[3 * i], self._conv_model[3 * i + 1]
            # (an inner layer's fp32 activation has no reader when the next layer gathers from its planes forward AND backward: decided by
            # the next layer's own predicates, ops_conv_planes.consumer_reads_planes_only -- not by position)
            nxt = True if i == last else ('conv', self._cnn_kernels[i + 1])
            x = conv2d(x, conv.weight, conv.bias, ln=(ln.norm.weight, ln.norm.bias, ln.norm.eps), fp32_out=nxt, planes_out=i != last)
        n, h, w, c = x.shape
        return ops.transpose_last2(x.reshape(n, h * w, c)).reshape(n, c * h * w)    # NCHW flatten, ref :621


class Decoder(Module):  # ref :631-715
    def __init__(self, shapes, cnn_keys=r'.*', mlp_keys=r'.*', act='SiLU', norm='none',
                 cnn_depth=48, cnn_kernels=(4, 4, 4, 4), mlp_layers=[400, 400, 400, 400], embed_dim=1024,
                 mlp_dist='mse', image_dist='mse'):
        super().__init__()
        self._embed_dim, self._shapes = embed_dim, shapes
        self.cnn_keys = [k for k, v in shapes.items() if re.match(cnn_keys, k) and len(v) == 3]
        self.mlp_keys = [k for k, v in shapes.items() if re.match(mlp_keys, k) and len(v) == 1]
        assert act == 'SiLU' and norm == 'layer' and image_dist == 'mse'
        assert len(self.mlp_keys) == 0 and len(self.cnn_keys) == 1
        self._cnn_depth, self._cnn_kernels = cnn_depth, cnn_kernels
        self.channels = {k: self._shapes[k][0] for k in self.cnn_keys}
        self._conv_in = nn.Sequential(nn.Linear(embed_dim, 32 * self._cnn_depth))
        layers, n = [], len(self._cnn_kernels)
        for i, kernel in enumerate(self._cnn_kernels):
            prev_depth = 32 * self._cnn_depth if i == 0 else 2 ** (n - (i - 1) - 2) * self._cnn_depth
            depth = 2 ** (n - i - 2) * self._cnn_depth
            last = i == n - 1
            if last:
                depth = sum(self.channels.values())
            layers += [nn.ConvTranspose2d(prev_depth, depth, kernel, stride=2),
                       NormLayer('none', depth) if last else ImgChLayerNorm(depth),
                       nn.Identity() if last else get_act(act)]
        self._conv_model = nn.Sequential(*layers)

    def forward(self, features):
        return self._cnn(features)

    def _cnn(self, features):
        lead = features.shape[:-1]
        x = ops.linear(features.reshape(-1, features.shape[-1]), self._conv_in[0].weight, self._conv_in[0].bias)
        x = x.reshape(-1, 1, 1, 32 * self._cnn_depth)             # NHWC with 1x1 pixels
        n = len(self._cnn_kernels)
        convT2d = ops_conv_planes.convT2d_s2 if ops_conv_planes.active(x) else ops.convT2d_s2
        for i in range(n):
            conv = self._conv_model[3 * i]
            if i != n - 1:
                ln = self._conv_model[3 * i + 1]
                # (the 3-channel last layer reads fp32 values, no planes; an inner layer's fp32 output is skipped when the next layer's own
                # predicates say it reads planes only)
                nconv = self._conv_model[3 * (i + 1)]
                nxt = True if i == n - 2 else ('convT', nconv.out_channels, self._cnn_kernels[i + 1])
                x = convT2d(x, conv.weight, conv.bias, ln=(ln.norm.weight, ln.norm.bias, ln.norm.eps), fp32_out=nxt, planes_out=i != n - 2)
            else:
                x = ops.convT2d_s2(x, conv.weight, conv.bias, out_nchw=True)     # frames leave in the reference's NCHW
        return {key: MSEDist(x.reshape(tuple(lead) + tuple(x.shape[1:]))) for key in self.channels}


# ----------------------------------------------------------------------------- RSSM

def _scan_noise(site, step_site, T, rows, K, dev):
    """Exp(1) noise of a whole scan, (T, rows, K).  Tests that replay the reference's per-step draws inject them under the step
    site's name (one tensor per obs_step call, `rssm.post` / `rssm.prior`): those are stacked in call order."""
    inj = noise._injected
    if inj is not None and site not in inj and step_site in inj:
        return torch.stack([noise.draw('exp', step_site, (rows, K), dev) for _ in range(T)], 0)
    return noise.draw('exp', site, (T, rows, K), dev)



class EnsembleRSSM(Module):  # ref :302-555
    def __init__(self, ensemble=5, stoch=30, deter=200, hidden=200, discrete=False, act='SiLU', norm='none',
                 std_act='softplus', min_std=0.1, action_dim=None, embed_dim=1536, device='cuda',
                 single_obs_posterior=False, cell_input='stoch', cell_type='gru'):
        super().__init__()
        assert action_dim is not None
        assert discrete and ensemble == 1 and cell_type == 'gru' and cell_input == 'stoch' and norm == 'layer', \
            'GenRL path: discrete latents, ensemble 1, GRU, layer norm'
        self.device = device
        self._embed_dim, self._action_dim, self._ensemble = embed_dim, action_dim, ensemble
        self._stoch, self._deter, self._hidden, self._discrete = stoch, deter, hidden, discrete
        self._norm, self._cell_type, self.cell_input = norm, cell_type, cell_input
        self.single_obs_posterior = single_obs_posterior
        self._cell = GRUCell(self._hidden, self._deter, norm=True, device=self.device)
        self._ensemble_img_dist = nn.ModuleList([nn.Linear(hidden, stoch * discrete) for _ in range(ensemble)])
        self._obs_dist = nn.Linear(hidden, stoch * discrete)
        self._img_in = nn.Sequential(nn.Linear(stoch * discrete + action_dim, hidden), NormLayer(norm, hidden))
        self._ensemble_img_out = nn.ModuleList(
            [nn.Sequential(nn.Linear(deter, hidden), NormLayer(norm, hidden)) for _ in range(ensemble)])
        in_obs = embed_dim if single_obs_posterior else deter + embed_dim
        self._obs_out = nn.Sequential(nn.Linear(in_obs, hidden), NormLayer(norm, hidden))

    # ---- shapes / helpers
    def initial(self, batch_size):
        z = lambda *s: _zeros(tuple(s), self.device)        # (read-only constants, built once per shape)
        return dict(logit=z(batch_size, self._stoch, self._discrete), stoch=z(batch_size, self._stoch, self._discrete),
                    deter=self._cell.get_initial_state(None, batch_size))

    def get_stoch_size(self):
        return self._stoch * self._discrete

    def get_deter_size(self):
        return self._cell.state_size

    def get_feat_size(self):
        return self.get_deter_size() + self.get_stoch_size()

    def get_stoch(self, state):
        s = state['stoch']
        return s.reshape(list(s.shape[:-2]) + [self._stoch * self._discrete])

    def get_deter(self, state):
        return state['deter']

    def get_feat(self, state):
        return torch.cat([self.get_stoch(state), self.get_deter(state)], -1)

    def get_dist(self, state, ensemble=False):
        assert not ensemble
        return OneHotDist(state['logit'].float())

    def get_unif_dist(self, state):
        return OneHotDist(torch.ones_like(state['logit']), site='rssm.unif')

    # ---- single steps (API parity; acting / data-free paths)
    def _prior_logits(self, deter):
        x = _dense_ln_silu(deter, self._ensemble_img_out[0][0], self._ensemble_img_out[0][1])
        lg = ops.linear(x, self._ensemble_img_dist[0].weight, self._ensemble_img_dist[0].bias)
        return lg.reshape(list(lg.shape[:-1]) + [self._stoch, self._discrete])

    def _post_logits(self, embed, deter=None):
        if self.single_obs_posterior:
            x = _dense_ln_silu(embed, self._obs_out[0], self._obs_out[1])
        else:
            x = _dense_ln_silu(deter, self._obs_out[0], self._obs_out[1], embed)
        lg = ops.linear(x, self._obs_dist.weight, self._obs_dist.bias)
        return lg.reshape(list(lg.shape[:-1]) + [self._stoch, self._discrete])

    def get_stoch_stats_from_deter_state(self, temp_state, sample=True, site='rssm.prior'):
        logit = self._prior_logits(temp_state['deter'])
        d = OneHotDist(logit, site=site)
        return (d.sample() if sample else d.mode()), {'logit': logit}

    def img_step(self, prev_state, prev_action, sample=True, site='rssm.prior'):
        x = _dense_ln_silu(self.get_stoch(prev_state), self._img_in[0], self._img_in[1], prev_action)
        deter = ops.gru_step(x, prev_state['deter'], self._cell._layer.weight, self._cell._norm.weight,
                             self._cell._norm.bias)
        stoch, stats = self.get_stoch_stats_from_deter_state({'deter': deter}, bool(sample), site)
        return {'stoch': stoch, 'deter': deter, **stats}

    def get_post_stoch(self, embed, prior, should_sample=True):
        logit = self._post_logits(embed, prior['deter'])
        d = OneHotDist(logit, site='rssm.post')
        return (d.sample() if should_sample else d.mode()), {'logit': logit}

    def obs_step(self, prev_state, prev_action, embed, is_first, should_sample=True):
        m = 1.0 - is_first.float()
        prev_state = {k: torch.einsum('b,b...->b...', m, v) for k, v in prev_state.items()}
        prev_action = torch.einsum('b,b...->b...', m, prev_action)
        prior = self.img_step(prev_state, prev_action, should_sample)
        stoch, stats = self.get_post_stoch(embed, prior, should_sample)
        return {'stoch': stoch, 'deter': prior['deter'], **stats}, prior

    # ---- sequences
    def observe(self, embed, action, is_first, state=None):
        """ref :362-371.  With single_obs_posterior the posterior, `_img_in`, the x-half of the GRU
        projection and the prior head are batched over all T steps; only h_{t-1} W_h + LN + gates is
        sequential (ops.gru_seq).  Without it, falls back to the step-by-step form."""
        B, T = action.shape[:2]
        if not self.single_obs_posterior:
            if os.environ.get('GENRL_OBSERVE_SEQ', '1') == '0':
                return self._observe_stepwise(embed, action, is_first, state)
            return self._observe_scan(embed, action, is_first, state)
        S, K = self._stoch, self._discrete
        dev = embed.device
        tm = lambda x: x.transpose(0, 1).contiguous()                          # (B,T,..) -> (T,B,..)
        emb, act, first = tm(embed), tm(action), tm(is_first)
        mask = (1.0 - first.float()).contiguous()                              # (T,B)
        plog = self._post_logits(emb.reshape(T * B, -1)).reshape(T, B, S, K)
        pst = ops.onehot_sample(plog, noise.draw('exp', 'wm.post_q', (T, B * S, K), dev))
        st0 = state if state is not None else self.initial(B)
        q_prior = noise.draw('exp', 'wm.prior_q', (T, B * S, K), dev)
        # Everything below only feeds the KL term (with `decoder_inputs: stoch` the decoder and the
        # reward head need the posterior alone): when `fork_prior` is set it is enqueued on a side
        # stream, so this latency-bound T-step chain runs beside the conv-heavy decoder work; autograd
        # replays each branch's backward on its own stream.  The caller joins before using `prior`.
        ctx = streams.fork('scan') if getattr(self, 'fork_prior', False) else contextlib.nullcontext()
        with ctx:
            prev = torch.cat([st0['stoch'].reshape(1, B, S * K), pst.reshape(T, B, S * K)[:-1]], 0)
            mrow = mask.reshape(T * B, 1)
            x = _dense_ln_silu((prev.reshape(T * B, S * K) * mrow), self._img_in[0], self._img_in[1],
                               act.reshape(T * B, -1) * mrow)
            deter = ops.gru_seq(x.reshape(T, B, -1), mask, st0['deter'], self._cell._layer.weight,
                                self._cell._norm.weight, self._cell._norm.bias)
            qlog = self._prior_logits(deter.reshape(T * B, -1)).reshape(T, B, S, K)
            qst = ops.onehot_sample(qlog, q_prior)
        bm = lambda x: x.transpose(0, 1)
        post = {'stoch': bm(pst), 'deter': bm(deter), 'logit': bm(plog)}
        prior = {'stoch': bm(qst), 'deter': bm(deter), 'logit': bm(qlog)}
        return post, prior

    def _observe_scan(self, embed, action, is_first, state=None):
        """ref :362-371 with the posterior on [deter, embed] (`single_obs_posterior: false`, conf/defaults/dreamer_v3.yaml:5): the
        sampled latent is inside the recurrence.  ONE autograd node (ops.observe_seq): the action half of `_img_in` and the embed half
        of `_obs_out` are batched over T, the remaining chain is eight launches per step each way from C (csrc/seq.hip); the prior head
        does not feed the recurrence and runs on all T * B rows afterwards."""
        B, T = action.shape[:2]
        S, K = self._stoch, self._discrete
        dev = embed.device
        tm = lambda x: x.transpose(0, 1).contiguous()                          # (B,T,..) -> (T,B,..)
        emb, act, first = tm(embed), tm(action), tm(is_first)
        mask = (1.0 - first.float()).contiguous()                              # (T,B)
        st0 = state if state is not None else self.initial(B)
        q_post = _scan_noise('wm.post_q', 'rssm.post', T, B * S, K, dev)
        q_prior = _scan_noise('wm.prior_q', 'rssm.prior', T, B * S, K, dev)
        lin_i, ln_i = self._img_in[0], self._img_in[1]._layer
        lin_o, ln_o = self._obs_out[0], self._obs_out[1]._layer
        deter, plog, pst = ops.observe_seq(
            emb, act, mask, st0['stoch'].reshape(B, S * K), st0['deter'], q_post,
            lin_i.weight, lin_i.bias, ln_i.weight, ln_i.bias, self._cell._layer.weight, self._cell._norm.weight, self._cell._norm.bias,
            lin_o.weight, lin_o.bias, ln_o.weight, ln_o.bias, self._obs_dist.weight, self._obs_dist.bias, ln_i.eps, ln_o.eps)
        qlog = self._prior_logits(deter.reshape(T * B, -1)).reshape(T, B, S, K)
        qst = ops.onehot_sample(qlog, q_prior)
        bm = lambda x: x.transpose(0, 1)
        post = {'stoch': bm(pst.reshape(T, B, S, K)), 'deter': bm(deter), 'logit': bm(plog.reshape(T, B, S, K))}
        prior = {'stoch': bm(qst), 'deter': bm(deter), 'logit': bm(qlog)}
        return post, prior

    def _observe_stepwise(self, embed, action, is_first, state=None):
        B, T = action.shape[:2]
        state = state if state is not None else self.initial(B)
        posts, priors = [], []
        for t in range(T):
            post, prior = self.obs_step(state, action[:, t], embed[:, t], is_first[:, t])
            posts.append(post); priors.append(prior); state = post
        st = lambda L: {k: torch.stack([d[k] for d in L], 1) for k in L[0]}
        return st(posts), st(priors)

    def imagine(self, action, state=None, sample=True):
        """ref :373-381: prior rollout for given actions (B,T,A)."""
        B, T = action.shape[:2]
        state = state if state is not None else self.initial(B)
        if not torch.is_grad_enabled() and action.is_cuda and os.environ.get('GENRL_OBSERVE_SEQ', '1') != '0':
            # forward only (the data-free block's warm-up rollouts, report, video_imagine all run under no_grad): the action half of
            # `_img_in` batched over T, the eight launches per step of the remaining chain from ONE host call (csrc/seq.hip)
            S, K = self._stoch, self._discrete
            lin_i, ln_i = self._img_in[0], self._img_in[1]._layer
            lin_o, ln_o = self._ensemble_img_out[0][0], self._ensemble_img_out[0][1]._layer
            q = _scan_noise('rssm.imagine_q', 'rssm.prior', T, B * S, K, action.device) if sample else None
            deter, logit, stoch = ops.rssm_imagine_seq(
                action.transpose(0, 1), state['stoch'].reshape(B, S * K), state['deter'], q, S, K,
                lin_i.weight, lin_i.bias, ln_i.weight, ln_i.bias, self._cell._layer.weight, self._cell._norm.weight,
                self._cell._norm.bias, lin_o.weight, lin_o.bias, ln_o.weight, ln_o.bias, self._ensemble_img_dist[0].weight,
                self._ensemble_img_dist[0].bias, ln_i.eps, ln_o.eps)
            bm = lambda x: x.transpose(0, 1)
            return {'stoch': bm(stoch.reshape(T, B, S, K)), 'deter': bm(deter), 'logit': bm(logit.reshape(T, B, S, K))}
        outs = []
        for t in range(T):
            state = self.img_step(state, action[:, t], sample)
            outs.append(state)
        return {k: torch.stack([d[k] for d in outs], 1) for k in outs[0]}

    # ---- losses
    def kl_loss(self, post, prior, forward, balance, free, free_avg):
        """ref :534-555 (balance != 0.5, free_avg False)."""
        assert balance != 0.5 and not free_avg
        lhs, rhs = (prior, post) if forward else (post, prior)
        mix = balance if forward else (1 - balance)
        l, r = lhs['logit'], rhs['logit']
        # mix * max(KL(l || sg r), free).mean() + (1 - mix) * max(KL(sg l || r), free).mean() as one autograd node
        return ops.kl_balance(l, r, mix, free)


# ----------------------------------------------------------------------------- optimiser

class FlatGroup:
    """Parameters of one optimiser group re-homed into one flat fp32 buffer (params and grads are
    views), so clip-norm + decay + Adam is one pass and DP needs one all-reduce per group."""
    ALIGN = 64          # floats: every parameter starts on a 256-byte boundary (the vector-load GEMMs need 16)
    NORM_SLOTS = 8

    def __init__(self, params):
        self.params = [p for p in params]
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.n = n
        self.flat = torch.zeros(n, device=dev)              # (the gaps stay zero: zero gradient, zero Adam moments)
        self.grad = torch.zeros(n, device=dev)
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.step = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)   # device-side Adam step (graph replay)
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
        # gradient-norm metric slots, used round-robin: the 0-d tensor an optimiser call hands out stays valid for the next
        # NORM_SLOTS - 1 steps of this group (a caller that aggregates metrics over a few steps reads what it was given)
        self.norm = torch.zeros(self.NORM_SLOTS, device=dev)
        self.norm_i = 0

    def norm_slot(self):
        k = self.norm_i
        self.norm_i = (k + 1) % self.NORM_SLOTS
        return self.norm[k:k + 1]

    def owns(self, params):
        return len(params) == len(self.params) and all(a is b for a, b in zip(params, self.params))

    def rebind(self):
        """re-attach .grad views (zero_grad(set_to_none) or external code may have dropped them)"""
        for p, off in zip(self.params, self.offsets):
            g = self.grad[off:off + p.numel()].view(p.shape)
            if p.grad is None:
                p.grad = g
            elif p.grad.data_ptr() != g.data_ptr():
                g.copy_(p.grad); p.grad = g


class Optimizer:
    """ref :871-932: backward -> clip by global norm -> p *= (1-wd) for *every* handed parameter
    -> Adam -> zero_grad, as HIP kernels over flat buffers and without host syncs (metrics are
    0-d device tensors).  `grad_reduce` is the DP hook (one all-reduce per group)."""
    grad_reduce = None      # set by genrl_amd.dp: callable(flat_grad) -> divisor
    grad_reduce_async = None   # set by genrl_amd.dp: callable(flat_grad) -> (wait(), world): the reduction runs beside later work
    overlap_under_dp = False   # set by genrl_amd.dp for RCCL: collectives are stream-ordered, the connector's side stream may stay on
    grad_hook = None        # test hook: callable(opt_name, params) after backward, before the step
    reduce_hook = None      # test hook: callable(opt_name, group, gscale) after the DP reduction, before clip / Adam

    def __init__(self, name, parameters, lr, eps=1e-4, clip=None, wd=None, opt='adam', wd_pattern=r'.*', use_amp=False):
        assert 0 <= wd < 1
        assert not clip or 1 <= clip
        assert opt == 'adam' and wd_pattern == r'.*'
        self._use_amp = use_amp   # precision 16 = bf16 MFMA operands (ops.set_gemm_precision): no GradScaler, nothing to do here
        self._name, self._clip, self._wd, self._lr, self._eps = name, clip, wd, lr, eps
        self._params = list(parameters)
        self._groups = []
        self._live_cache = {}
        self._pending = []
        self._once = True

    def _group_for(self, params):
        for g in self._groups:
            if g.owns(params):
                return g
        # a parameter can only live in one flat buffer: groups must be disjoint
        ids = {id(p) for p in params}
        for g in self._groups:
            assert not ids & {id(p) for p in g.params}, 'overlapping optimiser groups'
        g = FlatGroup(params)
        self._groups.append(g)
        return g

    def __call__(self, loss, params, decay_only=(), defer=False, flush_pending=True):
        """defer=True (data parallel only): the gradient all-reduce is STARTED here and the clip / Adam pass is left
        pending until flush() -- or this optimiser's next call, after that call's backward (flush_pending=False: not by
        that call -- it runs on a side stream and the pending step belongs to the main one): the reduction then runs
        beside whatever the caller enqueues in between (the connector update behind the world-model reduction, the
        critic update behind the actor's).  The returned grad-norm metric is a device scalar in one of the group's
        NORM_SLOTS round-robin slots, written when the step completes (stream order makes that invisible to a reader on
        the same stream): it keeps its value for the next NORM_SLOTS - 1 steps of the group; read (or clone) it before."""
        params = [p for p in params]
        assert len(loss.shape) == 0 or (len(loss.shape) == 1 and loss.shape[0] == 1), (self._name, loss.shape)
        metrics = {}
        if self._once:
            count = sum(p.numel() for p in params if p.requires_grad)
            print(f'Found {count} {self._name} parameters.')
            self._once = False
        metrics[f'{self._name}_loss'] = loss.detach()
        # parameters that get a gradient from this loss form the Adam group; the others handed in
        # only receive weight decay (connector weights during the world-model step).
        key = tuple(id(p) for p in params)
        if key not in self._live_cache:                 # graph topology is static across iterations
            self._live_cache[key] = [p for p in params if p.requires_grad and _in_graph(loss, p)]
        live = self._live_cache[key]
        group = self._group_for(live)
        group.rebind()
        ops.direct_grads = True        # weight-gradient kernels accumulate straight into the flat gradient buffers
        ops.defer_begin()              # ... and the LayerNorm backward passes leave their parameter-gradient partials unreduced
        try:
            loss.backward(gradient=_one_like(loss))       # (a cached seed: torch would fill a fresh ones tensor per call)
        except BaseException:
            ops.direct_grads = False
            ops.defer_abort()          # a failed backward pass: its partial sets (possibly never written) are NOT summed into the gradients
            raise
        ops.direct_grads = False
        ops.defer_flush()              # one launch sums them all (the backward pass's streams have been joined by autograd)
        ops.wgrad_stream.join()
        if Optimizer.grad_hook is not None:
            Optimizer.grad_hook(self._name, live)
        if flush_pending:
            self.flush()               # an earlier deferred step: its reduction had this backward to hide behind
        gscale, wait = 1.0, None
        if Optimizer.grad_reduce is not None:
            if defer and Optimizer.grad_reduce_async is not None:
                wait, world = Optimizer.grad_reduce_async(group.grad)
                gscale = 1.0 / world
            else:
                gscale = 1.0 / Optimizer.grad_reduce(group.grad)
        # weight decay of the handed-in parameters that are not live (disjoint from the Adam group: order-free)
        if self._wd:
            live_ids = {id(p) for p in live}
            for p in params:
                if id(p) not in live_ids:
                    other = self._flat_owner(p)
                    if other is None:
                        ops.scale_(p.data, 1.0 - self._wd) if p.data.is_contiguous() else p.data.mul_(1.0 - self._wd)
            for g in self._decay_groups(params, live_ids):
                ops.scale_(g.flat, 1.0 - self._wd)
            pl.invalidate([p for p in params if id(p) not in live_ids])     # (their cached weight planes are stale)
        slot = group.norm_slot()
        metrics[f'{self._name}_grad_norm'] = slot[0]
        pend = (group, gscale, wait, slot)
        if wait is not None:
            self._pending.append(pend)
        else:
            self._finish(pend)
        return metrics

    def _finish(self, pend):
        group, gscale, wait, slot = pend
        if wait is not None:
            wait()
        if Optimizer.reduce_hook is not None:
            Optimizer.reduce_hook(self._name, group, gscale)
        ops.grad_norm(group.grad, slot, gscale, step_inc=group.step_dev)     # (also: device step count += 1)
        group.step += 1
        pl.invalidate(group.params)         # (these weights change below: their cached weight planes are stale)
        ops.adam_step(group.flat, group.grad, group.m, group.v, slot, gscale, float(self._clip or 0.0),
                      self._lr, self._eps, float(self._wd or 0.0), group.step, step_dev=group.step_dev, zero_grad=True)
        # (the gradient buffer was cleared by the Adam pass)

    def flush(self):
        """complete the deferred steps (no-op otherwise)"""
        while self._pending:
            self._finish(self._pending.pop(0))

    def _flat_owner(self, p):
        for g in self._groups:
            for q in g.params:
                if q is p:
                    return g
        return None

    def _decay_groups(self, params, live_ids):
        """flat groups all of whose parameters were handed in without being live"""
        ids = {id(p) for p in params} - live_ids
        return [g for g in self._groups if g.params and all(id(q) in ids for q in g.params)]


def _in_graph(loss, p):
    """True if parameter p is reachable from loss in the autograd graph."""
    cache = getattr(loss, '_genrl_leafs', None)
    if cache is None:
        cache, seen, stack = set(), set(), [loss.grad_fn]
        while stack:
            fn = stack.pop()
            if fn is None or fn in seen:
                continue
            seen.add(fn)
            if hasattr(fn, 'variable'):
                cache.add(id(fn.variable))
            stack.extend(f for f, _ in fn.next_functions)
        try:
            loss._genrl_leafs = cache
        except Exception:
            pass
    return id(p) in cache


# ----------------------------------------------------------------------------- misc (ref :934-1029)

class StreamNorm:
    def __init__(self, shape=(), momentum=0.99, scale=1.0, eps=1e-8, device='cuda'):
        self.device, self._shape, self._momentum, self._scale, self._eps = device, tuple(shape), momentum, scale, eps
        self.mag = self.mean = self.square_mean = None
        self.step = 0

    def reset(self):
        self.step, self.mag, self.mean, self.square_mean = 0, None, None, None

    def __call__(self, inputs):
        """All seven reductions of the reference (:956-1001) from ONE pass over the inputs (ops.moments):
        [mean, std, mean |x|, mean x^2]."""
        metrics = {}
        scalar = self._shape == () and inputs.is_cuda
        mom = ops.moments(inputs) if scalar else None
        self.update(inputs, mom)
        metrics['mean'] = mom[0] if scalar else inputs.mean()
        metrics['std'] = mom[1] if scalar else inputs.std()
        outputs = self.transform(inputs)
        if self._momentum == 1 and scalar:                 # identity transform: the normed statistics are the same numbers
            metrics['normed_mean'], metrics['normed_std'] = mom[0], mom[1]
        else:
            metrics['normed_mean'] = outputs.mean()
            metrics['normed_std'] = outputs.std()
        return outputs, metrics

    def update(self, inputs, mom=None):
        self.step += 1
        if self._momentum == 1 and self.mag is not None:
            return                                  # ema(old, new) = 1 * old + 0 * new: the statistics never move again
        # in place once the state exists: under hipGraph replay the state tensors keep their capture-time addresses, so the EMA
        # advances across replays (a rebinding `old = m * old + (1 - m) * new` froze `old` at its warm-up value)
        def ema(old, new):
            if old is None:
                return new.detach().clone()
            return old.mul_(self._momentum).add_(new.detach(), alpha=1 - self._momentum)
        if mom is not None:        # scalar statistics from the one-pass kernel: the same EMA (ref :972-984), first call = copy
            self.mag, self.mean, self.square_mean = ema(self.mag, mom[2]), ema(self.mean, mom[0]), ema(self.square_mean, mom[3])
            return
        batch = inputs.detach().reshape((-1,) + self._shape)
        self.mag = ema(self.mag, torch.abs(batch).mean(0))
        self.mean = ema(self.mean, torch.mean(batch))
        self.square_mean = ema(self.square_mean, torch.mean(batch * batch))

    def transform(self, inputs):
        if self._momentum == 1:
            return inputs
        values = inputs.reshape((-1,) + self._shape) / (self.mag[None] + self._eps) * self._scale
        return values.reshape(inputs.shape)


class RequiresGrad:
    def __init__(self, model):
        self._model = model

    def __enter__(self):
        self._model.requires_grad_(requires_grad=True)

    def __exit__(self, *args):
        self._model.requires_grad_(requires_grad=False)


class RewardEMA:
    """ref :1014-1029.  `all_gather` is the DP hook so the quantiles see the global batch."""
    all_gather = None

    def __init__(self, device, alpha=1e-2):
        self.device, self.alpha = device, alpha
        self.range = torch.tensor([0.05, 0.95]).to(device)

    def __call__(self, x, ema_vals):
        """-> (offset, scale); also leaves self.last = tensor [offset, scale, q05, q95] (device)."""
        flat_x = torch.flatten(x.detach())
        if RewardEMA.all_gather is not None:
            flat_x = RewardEMA.all_gather(flat_x)
        if flat_x.is_cuda and ema_vals.is_contiguous():
            # radix-select quantiles + EMA + clip in one kernel (stats.hip) instead of sort + ~10 elementwise launches
            self.last = ops.quantile_ema(flat_x, ema_vals, self.alpha, 0.05, 0.95)
            return self.last[0], self.last[1]
        x_quantile = torch.quantile(input=flat_x, q=self.range)
        ema_vals[:] = self.alpha * x_quantile + (1 - self.alpha) * ema_vals
        scale = torch.clip(ema_vals[1] - ema_vals[0], min=1.0)
        self.last = None
        return ema_vals[0].detach(), scale.detach()
