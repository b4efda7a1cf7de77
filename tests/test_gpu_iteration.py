"""End-to-end GPU parity of one train.py iteration (update_wm incl. connector step, second
connector step, update_imag_behavior) through the drop-in modules, against
 (a) golden vectors generated from the reference (tests/golden/*.npz) and
 (b) the CPU oracle on the same weights / batch / injected noise.
fp32; losses within 1e-3 relative (north_star), most far tighter."""
import os
import numpy as np
import pytest
import torch

import detgen
from oracle import genrl_oracle as O
from oracle.iteration import run_iteration
from param_shapes import agent_param_shapes

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


class FakeClip:
    ignores_text = True

    def get_txt_feat(self, text):
        g = torch.Generator().manual_seed(123)
        return torch.nn.functional.normalize(torch.randn(1, 512, generator=g), dim=-1)


def sites_from(noise):
    return {'wm.post_q': noise['wm']['post_q'], 'wm.prior_q': noise['wm']['prior_q'],
            'conn.clip_eps': [noise['conn1']['clip_eps'], noise['conn2']['clip_eps']],
            'conn.init_q': [noise['conn1']['init_q'], noise['conn2']['init_q']],
            'conn.ikl_init_q': [noise['conn1']['ikl_init_q'], noise['conn2']['ikl_init_q']],
            'conn.ikl_step_q': [noise['conn1']['ikl_step_q'], noise['conn2']['ikl_step_q']],
            'imag.act_eps': noise['imag']['act_eps'], 'imag.step_q': noise['imag']['step_q'],
            'imag.target_init_q': noise['imag']['target_init_q']}


def run_product(name, lr_zero, cfg_over, ocfg_over, overlap=False):
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    from genrl_amd import config, noise as gnoise
    from genrl_amd.agent import dreamer_utils as common
    if isinstance(name, dict):           # synthetic configuration without a golden file: {'meta': (B,T,A,S,K,H,seed), 'img': px}
        g = {'meta': np.array(list(name['meta']) + [0]), 'img': np.array(name.get('img', 64))}
    else:
        g = dict(np.load(os.path.join(G, name)))
    B, T, A, S, K, H, seed, _ = [int(x) for x in g['meta']]
    over = dict(cfg_over)
    if lr_zero:
        for k in ('model_opt', 'actor_opt', 'critic_opt'):
            over[k] = dict(lr=0.0, wd=0.0)
    cfg = config.default_cfg(B, T, device='cuda', overlap_detached=overlap, **over)
    ag = config.make_agent(cfg, act_dim=A, img=int(g['img']) if 'img' in g else 64)
    ocfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H, **ocfg_over)
    p = detgen.det_state_dict(agent_param_shapes(ocfg), seed)
    ag.load_state_dict({k: v.cuda() for k, v in p.items()})
    ag.wm.viclip_model = FakeClip()
    batch_cpu = {k[len('batch.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('batch.')}
    if not batch_cpu:
        batch_cpu = {k: torch.from_numpy(v) for k, v in detgen.det_batch(B, T, A=A, img=int(g['img']), seed=seed).items()}
    batch = {k: v.cuda() for k, v in batch_cpu.items()}
    noise = detgen.iteration_noise(B, T, S, K, A, H, seed=seed)
    grads, phase = {}, []
    names = {id(q): n for n, q in ag.named_parameters()}

    def hook(opt_name, params):
        ph = {'model': 'wm' if 'wm' not in grads else ('conn1' if 'conn1' not in grads else 'conn2'),
              'actor': 'actor', 'critic': 'critic'}[opt_name]
        grads[ph] = {names[id(q)]: q.grad.detach().clone().cpu() for q in params}
    common.Optimizer.grad_hook = hook
    try:
        with gnoise.inject(sites_from(noise)):
            state, outputs, mets = ag.update_wm(batch, 0)
            mets_wm = {k: float(v) for k, v in mets.items()}
            _, mets = ag.wm.update_additional_detached_modules(batch, outputs, mets)
            _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics=mets, seq_data=batch)
    finally:
        common.Optimizer.grad_hook = None
    torch.cuda.synchronize()
    mets = {k: float(v) for k, v in mets.items()}
    return g, ocfg, p, batch_cpu, noise, ag, outputs, mets_wm, mets, grads


def measured(name, value):
    """GENRL_PARITY_REPORT=<file>: append the MEASURED value behind an assertion (the tolerances below are set to ~2x what this
    reports on MI355X, so that a regression of the plane arithmetic shows; profiles/r04_parity_measured.txt is one such report)"""
    path = os.environ.get('GENRL_PARITY_REPORT')
    if path:
        with open(path, 'a') as f:
            f.write(f'{name} {value:.3e}\n')


def check_vs_golden(g, mets_wm, mets, rtol):
    for key, val in g.items():
        for pre, src in (('metrics_wm.', mets_wm), ('metrics_conn2.', mets), ('metrics_imag.', mets)):
            if key.startswith(pre):
                n = key[len(pre):]
                if pre == 'metrics_wm.' and ('connector' in n or 'aligner' in n):
                    continue
                np.testing.assert_allclose(src[n], float(val), rtol=rtol, atol=2e-6, err_msg=n)


def test_tiny_iteration_vs_reference_and_oracle():
    tiny_o = dict(deter=32, hidden=32, units=32, cnn_depth=4)
    from genrl_amd import config
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product('tiny_iter.npz', True, config.tiny_overrides(), tiny_o)
    # sampled latent indices identical to the reference's
    assert (outputs['post']['stoch'].argmax(-1).cpu().numpy() == g['post_idx']).all()
    assert (outputs['prior']['stoch'].argmax(-1).cpu().numpy() == g['prior_idx']).all()
    assert (ag.unconditional_target['stoch'].argmax(-1).cpu().numpy() == g['target_idx']).all()
    c = lambda a, b, **kw: np.testing.assert_allclose(a.detach().cpu().numpy(), b, **kw)
    c(outputs['embed'], g['full.embed'], rtol=1e-4, atol=1e-5)
    c(outputs['post']['logit'], g['full.post_logit'], rtol=1e-4, atol=1e-5)
    c(outputs['prior']['logit'], g['full.prior_logit'], rtol=1e-4, atol=1e-5)
    c(outputs['kl'], g['full.kl'], rtol=1e-4, atol=1e-5)
    c(outputs['likes']['observation'], g['full.like_obs'], rtol=1e-5)
    c(outputs['likes']['reward'], g['full.like_rew'], rtol=1e-4, atol=1e-5)
    check_vs_golden(g, mets_wm, mets, 2e-4)
    # gradients against the reference's (stored for tensors <= 20k elements) and the oracle's
    n = 0
    for key, val in g.items():
        if key.startswith('grad.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(grads[ph][name].numpy(), val, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(val).max()), err_msg=key)
            n += 1
    assert n > 50
    text = FakeClip().get_txt_feat('')
    res = run_iteration(p, ocfg, batch, noise, text, apply_updates=False)
    for ph in ('wm', 'conn1', 'conn2', 'actor', 'critic'):
        assert set(grads[ph]) == set(res['grads'][ph]), (ph, set(grads[ph]) ^ set(res['grads'][ph]))
        for name, gref in res['grads'][ph].items():
            a, b = grads[ph][name].numpy(), gref.numpy()
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(b).max()), err_msg=f'{ph}.{name}')


def test_tiny_optimizer_step_vs_reference():
    tiny_o = dict(deter=32, hidden=32, units=32, cnn_depth=4)
    from genrl_amd import config
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product('tiny_opt.npz', False, config.tiny_overrides(), tiny_o)
    check_vs_golden(g, mets_wm, mets, 1e-3)
    sd = ag.state_dict()
    trained = {n for n, _ in ag.named_parameters()}
    bad, worst = 0, 0.0
    for key, val in g.items():
        if key.startswith('psum.'):
            name = key[len('psum.'):]
            if '_target_critic.' in name or name.startswith('_acting'):
                continue
            d = (sd[name].cpu() - p[name]).double()
            if not np.isclose(d.abs().sum().item(), val[1], rtol=0.05, atol=1e-7):
                bad += 1
            # SURVEY 8(c)'s elementwise bound, for THIS test too: one Adam step moves an element by <= lr (m / sqrt(v) = +-1 on the
            # first step) plus weight decay, so a parameter may differ from the reference's updated one by at most 2 lr per step
            # taken -- the connector's group steps twice per iteration (Q1)
            if name not in trained:                      # (buffers -- the return-quantile EMA -- are not optimiser state)
                continue
            lr = ag.cfg.model_opt.lr if name.startswith('wm.') else ag.cfg.actor_opt.lr
            nstep = 2 if name.startswith('wm.connector') else 1
            worst = max(worst, d.abs().max().item() / (lr * nstep))
            assert d.abs().max().item() <= 2.0 * lr * nstep * (1 + 1e-3), (name, d.abs().max().item(), lr, nstep)
    measured('tiny_opt.max_elem_delta_over_lr', worst)
    measured('tiny_opt.groups_outside_5pct_L1', bad)
    assert bad <= 3, bad
    # slow critic hard-copied after the first update (agent/dreamer.py:455-462)
    for k in sd:
        if k.startswith('_imag_behavior._target_critic.'):
            assert torch.equal(sd[k], sd[k.replace('_target_critic', 'critic')])


# tolerances of the full-width c1 case = ~2x the values measured on MI355X with plane operands forced on at every size
# (profiles/r04_parity_measured.txt): sampled-latent mismatches and the worst relative error of a per-tensor gradient L2 norm
# measured (round 4): 0 mismatching samples of 65 536; worst gradient-norm error 4.9e-7
C1_MAX_IDX_MISMATCH = 5e-5          # (three samples)
C1_GRAD_L2_RTOL = 5e-6


@pytest.mark.parametrize('overlap', [False, True])
def test_c1_full_dims_vs_reference(overlap):
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product('c1_full.npz', True, {}, {}, overlap=overlap)
    mism = (outputs['post']['stoch'].argmax(-1).cpu().numpy() != g['post_idx']).mean()
    measured(f'c1_full.post_idx_mismatch[overlap={overlap}]', mism)
    assert mism <= C1_MAX_IDX_MISMATCH, mism
    check_vs_golden(g, mets_wm, mets, 1e-3)
    worst = 0.0
    for key, val in g.items():
        if key.startswith('gsum.'):
            _, ph, name = key.split('.', 2)
            l2 = grads[ph][name].double().norm().item()
            worst = max(worst, abs(l2 - val[2]) / max(abs(val[2]), 1e-12) if abs(val[2]) > 1e-6 else 0.0)
            np.testing.assert_allclose(l2, val[2], rtol=C1_GRAD_L2_RTOL, atol=1e-6, err_msg=key)
    measured(f'c1_full.worst_grad_L2_rel[overlap={overlap}]', worst)


def test_c4_128px_five_layer_convs_vs_reference():
    """configs[3]-like: 128x128 frames, 5-layer encoder/decoder, kitchen A=9."""
    from genrl_amd import config
    over = config.tiny_overrides()
    over['encoder'] = dict(cnn_depth=4, cnn_kernels=[4, 4, 4, 4, 4]); over['decoder'] = dict(cnn_depth=4, cnn_kernels=[5, 5, 5, 6, 6])
    oc = dict(deter=32, hidden=32, units=32, cnn_depth=4, img=128, enc_kernels=(4, 4, 4, 4, 4), dec_kernels=(5, 5, 5, 6, 6))
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product('c4_tiny.npz', True, over, oc)
    assert outputs['embed'].shape[-1] == 256                 # 64 ch x 2 x 2
    assert (outputs['post']['stoch'].argmax(-1).cpu().numpy() == g['post_idx']).all()
    check_vs_golden(g, mets_wm, mets, 3e-4)
    for key, val in g.items():
        if key.startswith('gsum.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(grads[ph][name].double().norm().item(), val[2], rtol=2e-3, atol=1e-6, err_msg=key)


def test_c3_dreamer_agent_vs_reference_and_oracle():
    """configs[2]-like: DreamerAgent.update = update_wm + update_acting_behavior(env_reward) with
    dreamer_v3.yaml semantics (posterior from [deter, embed], decoder on feat, reward head trained,
    actor entropy bonus, T=18)."""
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    from genrl_amd import config, noise as gnoise
    from genrl_amd.agent import dreamer_utils as common
    from oracle.iteration import run_dreamer_iteration
    g = dict(np.load(os.path.join(G, 'c3_dreamer_tiny.npz')))
    B, T, A, S, K, H, seed, _ = [int(x) for x in g['meta']]
    zero = dict(lr=0.0, wd=0.0)
    cfg = config.dreamer_cfg(B, T, device='cuda', model_opt=zero, actor_opt=zero, critic_opt=zero,
                             **config.dreamer_tiny_overrides())
    ag = config.make_dreamer_agent(cfg, act_dim=A)
    ocfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H, deter=32, hidden=32, units=32, cnn_depth=4,
                      single_obs_posterior=False, decoder_inputs='feat', reward_grad=True, actor_ent=3e-4)
    p = detgen.det_state_dict(agent_param_shapes(ocfg, dreamer=True), seed)
    assert {k: tuple(v.shape) for k, v in ag.state_dict().items()} == {k: tuple(v.shape) for k, v in p.items()}
    ag.load_state_dict({k: v.cuda() for k, v in p.items()})
    bc = {k: torch.from_numpy(v) for k, v in detgen.det_batch(B, T, A=A, seed=seed).items() if k != 'clip_video'}
    batch = {k: v.cuda() for k, v in bc.items()}
    noise = detgen.iteration_noise(B, T, S, K, A, H, seed=seed)
    sites = {'rssm.prior': [noise['wm']['prior_q'][t] for t in range(T)],
             'rssm.post': [noise['wm']['post_q'][t] for t in range(T)],
             'imag.act_eps': noise['imag']['act_eps'], 'imag.step_q': noise['imag']['step_q']}
    grads = {}
    names = {id(q): n for n, q in ag.named_parameters()}
    common.Optimizer.grad_hook = lambda opt, params: grads.__setitem__(
        {'model': 'wm', 'actor': 'actor', 'critic': 'critic'}[opt], {names[id(q)]: q.grad.detach().clone().cpu() for q in params})
    try:
        with gnoise.inject(sites):
            state, outputs, mets = ag.update_wm(batch, 0)
            mets_wm = {k: float(v) for k, v in mets.items()}
            _, mets = ag.update_acting_behavior(state, outputs, {}, batch)
    finally:
        common.Optimizer.grad_hook = None
    mets = {k: float(v) for k, v in mets.items()}
    assert (outputs['post']['stoch'].argmax(-1).cpu().numpy() == g['post_idx']).all()
    for key, val in g.items():
        for pre, src in (('metrics_wm.', mets_wm), ('metrics_act.', mets)):
            if key.startswith(pre):
                np.testing.assert_allclose(src[key[len(pre):]], float(val), rtol=3e-4, atol=2e-6, err_msg=key)
    res = run_dreamer_iteration(p, ocfg, bc, noise)
    for ph in ('wm', 'actor', 'critic'):
        assert set(grads[ph]) == set(res['grads'][ph]), (ph, set(grads[ph]) ^ set(res['grads'][ph]))
        for name, gref in res['grads'][ph].items():
            a, b = grads[ph][name].numpy(), gref.numpy()
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(b).max()), err_msg=f'{ph}.{name}')


def test_c2_full_size_vs_oracle_and_determinism():
    """BASELINE configs[1] at its full size (B32 x T32, 64x64x3, H=16, A=10, default widths): every
    loss / metric of the iteration against the CPU oracle on the same weights, batch and noise
    (1e-3 relative, north_star), and bit-identical metrics when the same iteration is run twice
    (all reductions of the HIP path have a fixed order)."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    meta = {'meta': (32, 32, 10, 32, 32, 16, 3)}
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product(meta, True, {}, {})
    res = run_iteration(p, ocfg, batch, noise, FakeClip().get_txt_feat(''), apply_updates=False)
    om = {k: float(v) for k, v in res['metrics'].items()}
    checked = 0
    for k, v in mets.items():
        if k in om and np.isfinite(om[k]):
            np.testing.assert_allclose(v, om[k], rtol=1e-3, atol=1e-5, err_msg=k)
            checked += 1
    assert checked >= 20, checked
    for ph in ('wm', 'conn2', 'actor', 'critic'):      # whole-phase gradient norms
        a = np.sqrt(sum(float((t.double() ** 2).sum()) for t in grads[ph].values()))
        b = np.sqrt(sum(float((t.double() ** 2).sum()) for t in res['grads'][ph].values()))
        np.testing.assert_allclose(a, b, rtol=1e-3, err_msg=ph)
    _, _, _, _, _, _, _, mets_wm2, mets2, _ = run_product(meta, True, {}, {})
    assert mets_wm2 == mets_wm and mets2 == mets, 'HIP path is not run-to-run deterministic'


def _restore_fp32_arithmetic():
    """precision 16 switches process-wide state (GEMM mode, plane products): put the suite's defaults back"""
    from genrl_amd import ops, planes
    ops.set_gemm_precision(ops.F32_MODE)
    if planes._amp_saved is not None:
        planes.ENABLED, planes._amp_saved = planes._amp_saved, None


def test_precision16_vs_the_oracles_bf16_operand_mode():
    """cfg.precision = 16.  The PRODUCT's definition: every matrix product -- Linear / GRU / conv / transposed conv,
    forward, input gradient and weight gradient, the policy's output layer inside the head kernels included -- rounds both operands
    to bf16 (nearest even) and accumulates in fp32; LayerNorm, softmax, losses, optimiser and all tensors stay fp32; no GradScaler
    (bf16 has fp32's exponent range).  Pinned HERE against the oracle's restatement of exactly that arithmetic
    (`O.bf16_operands()`) on a full-width B4 x T16 iteration.

    What "pinned" can mean for this arithmetic: operands rounded to 8 bits turn a 1e-7 difference of summation order into a flipped
    rounding now and then, the flips into ~1e-3 differences a few layers on, and those into the odd flipped latent SAMPLE of the
    16-step rollout -- the oracle differs from ITSELF by that much when its rounded-operand products are accumulated in float64
    instead of float32 (measured: world-model phase 2e-6, imagination losses 1e-3, actor gradient norm 1e-2, 0.2 % of the imagined
    latents; a second host's BLAS moves the fp32-accumulated oracle by as much again).  So: world-model and connector metrics and
    gradient norms within north_star's 1e-3; every imagination-phase number within that self-noise -- 5e-3 (losses, statistics) /
    2e-2 (gradient norms) -- or three times the oracle's own fp32- vs fp64-accumulation spread of that number, whichever is larger.
    NOT pinned against the reference: its precision-16 path (fp16 autocast + GradScaler, agent/dreamer_utils.py:889-932) only runs
    on CUDA and could not be recorded -- "oracle-pinned, reference-unpinned"."""
    from genrl_amd import ops, planes
    meta = {'meta': (4, 16, 10, 32, 32, 16, 5), 'img': 64}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    try:
        g, ocfg, p, batch, noise, ag, outputs, w16, m16, grads = run_product(meta, True, dict(precision=16), {})
        assert ops.set_gemm_precision('bf16') == 'bf16' and not planes.ENABLED        # the mode was on, the fp32-grade plane products off
    finally:
        _restore_fp32_arithmetic()
    text = FakeClip().get_txt_feat('')
    with O.bf16_operands():
        res = run_iteration(p, ocfg, batch, noise, text, apply_updates=False)
    with O.bf16_operands(acc64=True):
        res64 = run_iteration(p, ocfg, batch, noise, text, apply_updates=False)
    om = {k: float(v) for k, v in res['metrics'].items()}
    om64 = {k: float(v) for k, v in res64['metrics'].items()}
    n, worst_wm, worst_imag = 0, 0.0, 0.0
    for k, v in {**w16, **m16}.items():
        if k not in om or not np.isfinite(om[k]):
            continue
        err = abs(v - om[k])
        if k.startswith('imag_'):
            floor = 2e-2 if k.endswith('grad_norm') else 5e-3           # (the mode's measured self-noise, see the docstring)
            tol = max(floor * abs(om[k]) + 1e-5, 3.0 * abs(om[k] - om64[k]))
            worst_imag = max(worst_imag, err / max(abs(om[k]), 1e-5))
        else:
            tol = 1e-3 * abs(om[k]) + 1e-5
            worst_wm = max(worst_wm, err / max(abs(om[k]), 1e-5))
        assert err <= tol, (k, v, om[k], om64[k])
        n += 1
    assert n >= 20, n
    measured('precision16.worst_wm_connector_metric_rel_vs_oracle', worst_wm)
    measured('precision16.worst_imagination_metric_rel_vs_oracle', worst_imag)
    # posterior latents (world-model phase): equal up to near-ties
    mism = (outputs['post']['stoch'].argmax(-1).cpu().numpy() != res['outs']['post']['stoch'].argmax(-1).numpy()).mean()
    measured('precision16.post_idx_mismatch', mism)
    assert mism < 2e-3, mism
    norm = lambda gs: np.sqrt(sum(float((t.double() ** 2).sum()) for t in gs.values()))
    for ph in ('wm', 'conn2', 'actor', 'critic'):
        a, b, b64 = norm(grads[ph]), norm(res['grads'][ph]), norm(res64['grads'][ph])
        tol = 1e-3 * b if ph in ('wm', 'conn2') else max(2e-2 * b, 3.0 * abs(b - b64))
        measured(f'precision16.grad_norm_rel_vs_oracle[{ph}]', abs(a - b) / b)
        assert abs(a - b) <= tol, (ph, a, b, b64)
    # ... and it is a DIFFERENT arithmetic from the fp32 path (the mode is really on), within bf16's error of it
    _, _, _, _, _, _, _, w32, m32, _ = run_product(meta, True, {}, {})
    assert planes.ENABLED and w16['model_loss'] != w32['model_loss']
    for k in ('model_loss', 'observation_loss', 'reward_loss', 'kl_loss', 'model_kl'):
        np.testing.assert_allclose(w16[k], w32[k], rtol=3e-2, err_msg=k)


@pytest.mark.parametrize('mode,planes_on', [('f32', False), ('bf16x3', True), ('bf16x3-big', False)])
def test_tiny_iteration_in_every_fp32_gemm_mode(mode, planes_on, monkeypatch):
    """The three fp32 arithmetic modes of the GEMM engine (GENRL_GEMM_MODE 0 / 3 / 2: fp32 MFMAs everywhere, the exact
    3-way bf16 split on every tile, the split on the 128x128 tile only) with and without the plane-operand (h2) rollout all
    reproduce the reference's tiny iteration: sampled latent indices exactly, metrics and gradients within the golden
    tolerances.  (The default -- mode 2 with plane operands -- is what every other test runs.)"""
    from genrl_amd import config, ops, planes
    monkeypatch.setattr(ops, 'F32_MODE', mode)
    monkeypatch.setattr(planes, 'ENABLED', planes_on)
    tiny_o = dict(deter=32, hidden=32, units=32, cnn_depth=4)
    try:
        g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product('tiny_iter.npz', True, config.tiny_overrides(), tiny_o)
        assert ops.set_gemm_precision(mode) == mode           # the agent's entry points selected it
    finally:
        ops.set_gemm_precision('bf16x3-big')
    assert (outputs['post']['stoch'].argmax(-1).cpu().numpy() == g['post_idx']).all()
    assert (ag.unconditional_target['stoch'].argmax(-1).cpu().numpy() == g['target_idx']).all()
    check_vs_golden(g, mets_wm, mets, 2e-4)
    n = 0
    for key, val in g.items():
        if key.startswith('grad.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(grads[ph][name].numpy(), val, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(val).max()), err_msg=key)
            n += 1
    assert n > 50
