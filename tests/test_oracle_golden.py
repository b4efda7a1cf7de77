"""Pin the oracle (oracle/genrl_oracle.py) against golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only."""
import os
import numpy as np
import pytest
import torch

import detgen
from oracle import genrl_oracle as O
from oracle.iteration import run_iteration

G = os.path.join(os.path.dirname(__file__), 'golden')
T_ = torch.from_numpy


def load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_kat_lambda_return():
    k = load('kat.npz')
    out = O.lambda_return(T_(k['lr_reward']), T_(k['lr_value']), T_(k['lr_disc']), T_(k['lr_boot']), 0.95)
    close(out, k['lr_out'])


def test_kat_twohot_symlog():
    k = load('kat.npz')
    close(O.twohot_logprob(T_(k['th_logits']), T_(k['th_x'])), k['th_logprob'])
    close(O.twohot_mean(T_(k['th_logits'])), k['th_mean'])
    close(O.symlog(T_(k['sl_x'])), k['sl_symlog']); close(O.symexp(T_(k['sl_x'])), k['sl_symexp'])


def test_kat_onehot():
    k = load('kat.npz')
    lp, lq = T_(k['oh_lp']), T_(k['oh_lq'])
    close(O.cat_kl(lp, lq), k['oh_kl']); close(O.cat_entropy(lp), k['oh_ent'])
    close(O.unimix_probs(lp), k['oh_probs']); close(O.onehot_mode(lp), k['oh_mode'])


def test_kat_gru_chln_maxcos():
    k = load('kat.npz')
    p = {'x.' + n[len('gru_p.'):]: T_(v) for n, v in k.items() if n.startswith('gru_p.')}
    p = {n.replace('x._layer', 'x._cell._layer').replace('x._norm', 'x._cell._norm'): v for n, v in p.items()}
    close(O.gru_cell(p, 'x.', T_(k['gru_x']), T_(k['gru_h'])), k['gru_out'])
    close(O.ch_layer_norm(T_(k['chln_x']), T_(k['chln_p.norm.weight']), T_(k['chln_p.norm.bias'])), k['chln_out'])
    close(O.max_cosine_similarity(T_(k['mc_u']), T_(k['mc_v'])), k['mc_out'])


def agent_shapes(cfg):
    """Parameter names/shapes of GenRLAgent.state_dict() (SURVEY §8a weight contract)."""
    from param_shapes import agent_param_shapes
    return agent_param_shapes(cfg)


def setup_case(name, **cfg_over):
    g = load(name)
    B, T, A, S, K, H, seed, lr_zero = [int(x) for x in g['meta']]
    cfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H, **cfg_over)
    p = detgen.det_state_dict(agent_shapes(cfg), seed)
    batch = {k[len('batch.'):]: T_(v) for k, v in g.items() if k.startswith('batch.')}
    if not batch:       # regenerated from its seed (tests/detgen.py)
        batch = {k: T_(v) for k, v in detgen.det_batch(B, T, A=A, img=int(g['img']), seed=seed).items()}
    noise = detgen.iteration_noise(B, T, S, K, A, H, seed=seed)
    gtxt = torch.Generator().manual_seed(123)
    text = torch.nn.functional.normalize(torch.randn(1, 512, generator=gtxt), dim=-1)
    return g, cfg, p, batch, noise, text


def summarize(t, k=8):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()] + t[:k].tolist())


def check_metrics(res, g, rtol):
    for key, val in g.items():
        for pre in ('metrics_wm.', 'metrics_conn2.', 'metrics_imag.'):
            if key.startswith(pre):
                n = key[len(pre):]
                if pre == 'metrics_wm.' and ('connector' in n or 'aligner' in n):
                    continue        # overwritten by the second connector update; compared via conn2
                np.testing.assert_allclose(float(res['metrics'][n]), float(val), rtol=rtol, atol=1e-6, err_msg=n)


def test_tiny_iteration_vs_reference():
    g, cfg, p, batch, noise, text = setup_case('tiny_iter.npz', deter=32, hidden=32, units=32, cnn_depth=4)
    res = run_iteration(p, cfg, batch, noise, text, apply_updates=False)
    assert (res['outs']['post']['stoch'].argmax(-1).numpy() == g['post_idx']).all()
    assert (res['outs']['prior']['stoch'].argmax(-1).numpy() == g['prior_idx']).all()
    assert (res['seq']['stoch'].argmax(-1).numpy() == g['imag_idx']).all()
    assert (res['target_cache']['stoch'].argmax(-1).numpy() == g['target_idx']).all()
    close(res['outs']['embed'], g['full.embed']); close(res['outs']['post']['logit'], g['full.post_logit'])
    close(res['outs']['prior']['logit'], g['full.prior_logit']); close(res['outs']['kl'], g['full.kl'])
    close(res['outs']['likes']['observation'], g['full.like_obs'], rtol=1e-5)
    close(res['outs']['likes']['reward'], g['full.like_rew'])
    close(res['seq']['feat'], g['full.imag_feat']); close(res['seq']['action'], g['full.imag_action'])
    close(res['reward'], g['full.reward']); close(res['lambda_target'], g['full.lambda_target'])
    check_metrics(res, g, 2e-5)
    n = 0
    for key, val in g.items():
        if key.startswith('grad.'):
            _, ph, name = key.split('.', 2)
            close(res['grads'][ph][name], val, rtol=2e-4, atol=2e-6); n += 1
        if key.startswith('gsum.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(summarize(res['grads'][ph][name], 4)[1:3], val[1:3], rtol=2e-4, atol=1e-5, err_msg=key)
    assert n > 50


def test_tiny_optimizer_step_vs_reference():
    g, cfg, p, batch, noise, text = setup_case('tiny_opt.npz', deter=32, hidden=32, units=32, cnn_depth=4)
    res = run_iteration(p, cfg, batch, noise, text, apply_updates=True)
    check_metrics(res, g, 5e-4)
    bad = 0
    for key, val in g.items():
        if key.startswith('psum.'):
            name = key[len('psum.'):]
            if name.endswith('_target_critic') or '_target_critic.' in name or name.startswith('_acting'):
                continue
            d = (res['p'][name] - p[name]).double()
            # Adam's first step is ~lr*sign(g): near-zero gradients flip sign under reassociation,
            # so compare the L1 size of the update, loosely, and the count of outliers.
            if not np.isclose(d.abs().sum().item(), val[1], rtol=0.05, atol=1e-7):
                bad += 1
    assert bad <= 3, bad


def test_c1_full_dims_vs_reference():
    g, cfg, p, batch, noise, text = setup_case('c1_full.npz')
    torch.set_num_threads(8)
    res = run_iteration(p, cfg, batch, noise, text, apply_updates=False)
    mism = (res['outs']['post']['stoch'].argmax(-1).numpy() != g['post_idx']).mean()
    assert mism < 1e-3
    check_metrics(res, g, 2e-4)
    for key, val in g.items():
        if key.startswith('gsum.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(summarize(res['grads'][ph][name], 4)[2], val[2], rtol=2e-3, atol=1e-6, err_msg=key)


C4 = dict(deter=32, hidden=32, units=32, cnn_depth=4, img=128, enc_kernels=(4, 4, 4, 4, 4), dec_kernels=(5, 5, 5, 6, 6))


def test_c4_128px_five_layer_convs_vs_reference():
    """configs[3]-like: 128x128 frames, 5-layer conv stacks, A=9, an is_first inside a window."""
    g, cfg, p, batch, noise, text = setup_case('c4_tiny.npz', **C4)
    res = run_iteration(p, cfg, batch, noise, text, apply_updates=False)
    assert (res['outs']['post']['stoch'].argmax(-1).numpy() == g['post_idx']).all()
    assert (res['seq']['stoch'].argmax(-1).numpy() == g['imag_idx']).all()
    check_metrics(res, g, 5e-5)
    for key, val in g.items():
        if key.startswith('gsum.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(summarize(res['grads'][ph][name], 4)[1:3], val[1:3], rtol=5e-4, atol=1e-5, err_msg=key)


def test_c3_dreamer_agent_vs_reference():
    """configs[2]-like: DreamerAgent + dreamer_v3.yaml (posterior from [deter, embed], decoder on feat,
    reward head trained end-to-end, env_reward, actor entropy 3e-4, T not a multiple of 8)."""
    from oracle.iteration import run_dreamer_iteration
    g = load('c3_dreamer_tiny.npz')
    B, T, A, S, K, H, seed, _ = [int(x) for x in g['meta']]
    cfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H, deter=32, hidden=32, units=32, cnn_depth=4,
                     single_obs_posterior=False, decoder_inputs='feat', reward_grad=True, actor_ent=3e-4)
    from param_shapes import agent_param_shapes
    p = detgen.det_state_dict(agent_param_shapes(cfg, dreamer=True), seed)
    batch = {k: T_(v) for k, v in detgen.det_batch(B, T, A=A, seed=seed).items()}
    noise = detgen.iteration_noise(B, T, S, K, A, H, seed=seed)
    res = run_dreamer_iteration(p, cfg, batch, noise)
    assert (res['outs']['post']['stoch'].argmax(-1).numpy() == g['post_idx']).all()
    for key, val in g.items():
        for pre in ('metrics_wm.', 'metrics_act.'):
            if key.startswith(pre):
                np.testing.assert_allclose(float(res['metrics'][key[len(pre):]]), float(val), rtol=5e-5, atol=1e-6, err_msg=key)
        if key.startswith('gsum.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(summarize(res['grads'][ph][name], 4)[1:3], val[1:3], rtol=5e-4, atol=1e-5, err_msg=key)


def test_bf16_operand_mode_of_the_oracle():
    """`with O.bf16_operands()` (the product's `precision: 16` arithmetic: every matrix product rounds both operands to bf16,
    fp32 accumulation; oracle-pinned only): the products are exactly the fp32 products of the rounded operands -- forward, input
    gradient and weight gradient -- and a whole tiny iteration runs, stays finite and tracks the fp32 one within bf16's error."""
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(5, 3, 24, generator=gen, requires_grad=True)
    w = torch.randn(16, 24, generator=gen, requires_grad=True)
    b = torch.randn(16, generator=gen, requires_grad=True)
    dy = torch.randn(5, 3, 16, generator=gen)
    r = lambda t: t.to(torch.bfloat16).float()
    with O.bf16_operands():
        y = O.F.linear(x, w, b)
        y.backward(dy)
    assert torch.equal(y.detach(), torch.nn.functional.linear(r(x), r(w), b).detach())
    assert torch.allclose(x.grad, r(dy) @ r(w), rtol=0, atol=1e-6)
    assert torch.allclose(w.grad, r(dy).reshape(-1, 16).t() @ r(x).reshape(-1, 24).detach(), rtol=0, atol=1e-5)
    assert torch.allclose(b.grad, dy.reshape(-1, 16).sum(0), rtol=0, atol=1e-6)
    for kind, shape_w in (('conv2d', (6, 4, 4, 4)), ('conv_transpose2d', (4, 6, 4, 4))):
        xi = torch.randn(2, 4, 10, 10, generator=gen, requires_grad=True)
        wi = torch.randn(*shape_w, generator=gen, requires_grad=True)
        with O.bf16_operands():
            yi = getattr(O.F, kind)(xi, wi, None, stride=2)
        ref = getattr(torch.nn.functional, kind)(r(xi), r(wi), None, stride=2)
        assert torch.equal(yi.detach(), ref.detach())
        gy = torch.randn(yi.shape, generator=gen)
        yi.backward(gy)
        xr, wr = r(xi).detach().requires_grad_(True), r(wi).detach().requires_grad_(True)
        gx, gw = torch.autograd.grad(getattr(torch.nn.functional, kind)(xr, wr, None, stride=2), (xr, wr), r(gy))
        assert torch.equal(xi.grad, gx) and torch.equal(wi.grad, gw)
    g, cfg, p, batch, noise, text = setup_case('tiny_iter.npz', deter=32, hidden=32, units=32, cnn_depth=4)
    res32 = run_iteration(p, cfg, batch, noise, text, apply_updates=False)
    with O.bf16_operands():
        res16 = run_iteration(p, cfg, batch, noise, text, apply_updates=False)
    assert not O._BF16_OPERANDS
    a, b_ = float(res16['metrics']['model_loss']), float(res32['metrics']['model_loss'])
    assert np.isfinite(a) and a != b_ and abs(a - b_) <= 3e-2 * abs(b_), (a, b_)
