import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch, torch.nn.functional as F
import test_gpu_iteration as T
from genrl_amd import config, ops
from oracle import genrl_oracle as O

orig = {k: getattr(ops, k) for k in ['maxcos', 'lambda_return', 'twohot_mean', 'actor_sample', 'gru_step', 'linear2', 'onehot_sample', 'ln_act', 'linear']}
def t_maxcos(u, v, urow=None):
    u2 = u.reshape(-1, u.shape[-1])[urow] if urow is not None else u
    return O.max_cosine_similarity(u2.detach(), v)
def t_lambda(reward, value, disc, lam):
    return O.lambda_return(reward, value[:-1], disc * torch.ones_like(reward), value[-1], lam)
def t_thmean(logits):
    b = torch.linspace(-20, 20, 255, device=logits.device)
    return O.symexp(torch.sum(torch.softmax(logits, -1) * b, -1, keepdim=True))
def t_actor(raw, eps, mn=0.1, mx=1.0):
    A = raw.shape[-1] // 2
    return torch.tanh(raw[..., :A]) + ((mx - mn) * torch.sigmoid(raw[..., A:] + 2.0) + mn) * eps
def t_gru(x, h, W, g, b):
    parts = F.layer_norm(F.linear(torch.cat([x, h], -1), W), (W.shape[0],), g, b, 1e-5)
    r, c, u = torch.chunk(parts, 3, -1)
    r = torch.sigmoid(r); c = torch.tanh(r * c); u = torch.sigmoid(u - 1.0)
    return u * c + (1 - u) * h
def t_lin2(x1, x2, W, b=None):
    return F.linear(torch.cat([x1, x2], -1), W, b)
def t_onehot(l, q): return O.onehot_sample(l, q)
def t_ln(x, g, b, eps=1e-5, act=True):
    y = F.layer_norm(x, (x.shape[-1],), g, b, eps); return F.silu(y) if act else y
def t_linear(x, W, b=None): return F.linear(x, W, b)
rep = dict(maxcos=t_maxcos, lambda_return=t_lambda, twohot_mean=t_thmean, actor_sample=t_actor, gru_step=t_gru, linear2=t_lin2, onehot_sample=t_onehot, ln_act=t_ln, linear=t_linear)
tiny_o = dict(deter=32, hidden=32, units=32, cnn_depth=4)
for name in [None] + list(rep):
    for k, v in orig.items(): setattr(ops, k, v)
    if name: setattr(ops, name, rep[name])
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = T.run_product('tiny_iter.npz', True, config.tiny_overrides(), tiny_o)
    print(f'{str(name):15s} actor_grad_norm {mets["imag_actor_grad_norm"]:.7f} (ref {float(g["metrics_imag.imag_actor_grad_norm"]):.7f})  conn {mets["connector_model_grad_norm"]:.7f}')
