"""Names and shapes of GenRLAgent.state_dict() (the weight contract of SURVEY.md §8a), derived
from the dims in an oracle cfg.  Checked against the reference in tests/test_oracle_golden.py when
/root/reference is present and against the product modules in tests/test_boundary.py."""


def _mlp(pre, inp, units, layers, out):
    d = {}
    for i in range(layers):
        d[f'{pre}dense{i}.weight'] = (units, inp if i == 0 else units)
        d[f'{pre}dense{i}.bias'] = (units,)
        d[f'{pre}norm{i}._layer.weight'] = (units,)
        d[f'{pre}norm{i}._layer.bias'] = (units,)
    for name, n in out:
        d[f'{pre}_out.{name}.weight'] = (n, units)
        d[f'{pre}_out.{name}.bias'] = (n,)
    return d


def _rssm(pre, c, action_dim, embed_dim=None, single=True):
    S = c.stoch * c.discrete
    d = {f'{pre}_cell._layer.weight': (3 * c.deter, c.hidden + c.deter),
         f'{pre}_cell._norm.weight': (3 * c.deter,), f'{pre}_cell._norm.bias': (3 * c.deter,),
         f'{pre}_ensemble_img_dist.0.weight': (S, c.hidden), f'{pre}_ensemble_img_dist.0.bias': (S,)}
    if embed_dim is not None:
        d[f'{pre}_obs_dist.weight'] = (S, c.hidden); d[f'{pre}_obs_dist.bias'] = (S,)
    d.update({f'{pre}_img_in.0.weight': (c.hidden, S + action_dim), f'{pre}_img_in.0.bias': (c.hidden,),
              f'{pre}_img_in.1._layer.weight': (c.hidden,), f'{pre}_img_in.1._layer.bias': (c.hidden,),
              f'{pre}_ensemble_img_out.0.0.weight': (c.hidden, c.deter), f'{pre}_ensemble_img_out.0.0.bias': (c.hidden,),
              f'{pre}_ensemble_img_out.0.1._layer.weight': (c.hidden,), f'{pre}_ensemble_img_out.0.1._layer.bias': (c.hidden,)})
    if embed_dim is not None:
        d.update({f'{pre}_obs_out.0.weight': (c.hidden, embed_dim if single else embed_dim + c.deter), f'{pre}_obs_out.0.bias': (c.hidden,),
                  f'{pre}_obs_out.1._layer.weight': (c.hidden,), f'{pre}_obs_out.1._layer.bias': (c.hidden,)})
    return d


def conv_out_sizes(img, kernels):
    s, out = img, []
    for k in kernels:
        s = (s - k) // 2 + 1
        out.append(s)
    return out


def embed_dim(c):
    sizes = conv_out_sizes(c.img, c.enc_kernels)
    return (2 ** (len(c.enc_kernels) - 1)) * c.cnn_depth * sizes[-1] ** 2


def agent_param_shapes(c, dreamer=False):
    """dreamer=True: DreamerAgent (no connector / _imag_behavior; decoder may take feat)."""
    d = {}
    n = len(c.enc_kernels)
    for i, k in enumerate(c.enc_kernels):
        cin = 3 if i == 0 else 2 ** (i - 1) * c.cnn_depth
        co = 2 ** i * c.cnn_depth
        d[f'wm.encoder._conv_model.{3*i}.weight'] = (co, cin, k, k)
        d[f'wm.encoder._conv_model.{3*i}.bias'] = (co,)
        d[f'wm.encoder._conv_model.{3*i+1}.norm.weight'] = (co,)
        d[f'wm.encoder._conv_model.{3*i+1}.norm.bias'] = (co,)
    E = embed_dim(c)
    S = c.stoch * c.discrete
    F = S + c.deter
    d.update(_rssm('wm.rssm.', c, c.act_dim, E, getattr(c, 'single_obs_posterior', True)))
    d['wm.heads.decoder._conv_in.0.weight'] = (32 * c.cnn_depth, S if getattr(c, 'decoder_inputs', 'stoch') == 'stoch' else F)
    d['wm.heads.decoder._conv_in.0.bias'] = (32 * c.cnn_depth,)
    n = len(c.dec_kernels)
    for i, k in enumerate(c.dec_kernels):
        cin = 32 * c.cnn_depth if i == 0 else 2 ** (n - (i - 1) - 2) * c.cnn_depth
        co = 3 if i == n - 1 else 2 ** (n - i - 2) * c.cnn_depth
        d[f'wm.heads.decoder._conv_model.{3*i}.weight'] = (cin, co, k, k)
        d[f'wm.heads.decoder._conv_model.{3*i}.bias'] = (co,)
        if i != n - 1:
            d[f'wm.heads.decoder._conv_model.{3*i+1}.norm.weight'] = (co,)
            d[f'wm.heads.decoder._conv_model.{3*i+1}.norm.bias'] = (co,)
    d.update(_mlp('wm.heads.reward.', F, c.units, c.mlp_layers, [('_out', 255)]))
    if dreamer:
        return _behaviors(d, c, F, ('_acting_behavior.',))
    ca = c.clip_dim + c.n_frames
    d.update(_rssm('wm.connector.', c, ca, None))
    al, D, M = 'wm.connector.aligner.', c.clip_dim, c.clip_dim // 2

    def res(pre, i, o, norm=True):
        if norm:
            d[f'{pre}norm_layer._layer.weight'] = (o,); d[f'{pre}norm_layer._layer.bias'] = (o,)
        d[f'{pre}layer.weight'] = (o, i); d[f'{pre}layer.bias'] = (o,)
        if i != o:
            d[f'{pre}res_proj.weight'] = (o, i); d[f'{pre}res_proj.bias'] = (o,)
    res(f'{al}down.0.', D, D); res(f'{al}down.1.', D, M); res(f'{al}mid.0.', M, M); res(f'{al}mid.1.', M, M)
    res(f'{al}up.0.', 2 * M, D, norm=False); res(f'{al}up.1.', 2 * D, D)
    ip = 'wm.connector.initial_state_pred.'
    d.update({f'{ip}0.weight': (c.hidden, ca), f'{ip}0.bias': (c.hidden,),
              f'{ip}1._layer.weight': (c.hidden,), f'{ip}1._layer.bias': (c.hidden,),
              f'{ip}3.weight': (c.hidden, c.hidden), f'{ip}3.bias': (c.hidden,),
              f'{ip}4._layer.weight': (c.hidden,), f'{ip}4._layer.bias': (c.hidden,),
              f'{ip}6.weight': (c.deter, c.hidden), f'{ip}6.bias': (c.deter,)})
    return _behaviors(d, c, F, ('_acting_behavior.', '_imag_behavior.'))


def _behaviors(d, c, F, names):
    for b in names:
        d[f'{b}ema_vals'] = (2,)
        d.update(_mlp(f'{b}actor.', F, c.units, c.mlp_layers, [('_out', c.act_dim), ('_std', c.act_dim)]))
        d.update(_mlp(f'{b}critic.', F, c.units, c.mlp_layers, [('_out', 255)]))
        d.update(_mlp(f'{b}_target_critic.', F, c.units, c.mlp_layers, [('_out', 255)]))
    return d
