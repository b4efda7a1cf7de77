"""Dense -> LayerNorm -> SiLU in ONE launch (genrl_gemm_h2_ln, csrc/gemm_planes.hip LnEpi; agent/dreamer_utils.py:718-747 MLP layers,
:459-473 img_step): the column tiles of a 64-row block exchange their partial row statistics inside one XCD's L2 behind a barrier of
N / 64 workgroups.  Against the two launches it replaces (genrl_gemm_h2 + genrl_ln_act_fwd_h2) and against float64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def _case(M, N, K0, K1, seed=0):
    from genrl_amd import planes
    A0 = torch.randn(M, K0, generator=g(seed + 1)).cuda() * torch.exp(torch.randn(M, 1, generator=g(seed + 2))).cuda()
    B0 = (torch.randn(N, K0, generator=g(seed + 3)) / (K0 + K1) ** .5).cuda()
    A1 = torch.randn(M, K1, generator=g(seed + 4)).cuda() * 3 if K1 else None
    B1 = (torch.randn(N, K1, generator=g(seed + 5)) / (K0 + K1) ** .5).cuda() if K1 else None
    bias = (0.3 * torch.randn(N, generator=g(seed + 6))).cuda()
    gamma = (1 + 0.2 * torch.randn(N, generator=g(seed + 7))).cuda()
    beta = (0.2 * torch.randn(N, generator=g(seed + 8))).cuda()
    P = [planes.split(A0), planes.split(B0), planes.split(A1) if K1 else None, planes.split(B1) if K1 else None]
    return P, bias, gamma, beta


SHAPES = [(1024, 1024, 1024, 0), (1024, 1024, 1024, 1024), (128, 1024, 1024, 64), (200, 512, 640, 0), (64, 64, 64, 0), (1000, 1024, 1024, 64),
          (512, 256, 128, 0), (1024, 960, 512, 0)]


@pytest.mark.parametrize('M,N,K0,K1', SHAPES)
def test_fused_product_layernorm_against_the_two_launches_and_float64(M, N, K0, K1):
    from genrl_amd import planes, ops_planes
    from genrl_amd._lib import lib
    assert lib().genrl_gemm_h2_ln_ok(M, N) == 1
    (a0, b0, a1, b1), bias, gamma, beta = _case(M, N, K0, K1)
    eps = 1e-3
    # the two launches
    C_ref = torch.empty(M, N, device='cuda'); y_ref = torch.empty(M, N, device='cuda')
    m_ref = torch.empty(M, device='cuda'); r_ref = torch.empty(M, device='cuda')
    p_ref = planes.Planes(M, N, 'cuda')
    planes.gemm(a0, b0, C_ref, N, bias, M, N, A1=a1, B1=b1)
    ops_planes._ln_fwd(C_ref.data_ptr(), gamma, beta, y_ref.data_ptr(), m_ref.data_ptr(), r_ref.data_ptr(), M, N, eps, p_ref, 0)
    # one launch, three times over (the barrier counters re-arm themselves)
    for rep in range(3):
        C = torch.full((M, N), float('nan'), device='cuda'); y = torch.full((M, N), float('nan'), device='cuda')
        mean = torch.full((M,), float('nan'), device='cuda'); rstd = torch.full((M,), float('nan'), device='cuda')
        out_p = planes.Planes(M, N, 'cuda')
        out_p.inv.fill_(float('nan'))
        assert planes.gemm_ln_ok(M, N)
        planes.gemm_ln(a0, b0, C, bias, M, N, gamma, beta, eps, out_p, 0, y=y, mean=mean, rstd=rstd, A1=a1, B1=b1)
        torch.cuda.synchronize()
        planes.check_ln_failure()
        assert torch.equal(C, C_ref)                                   # the same product, bit for bit
        ref = F.silu(F.layer_norm(C.double(), (N,), gamma.double(), beta.double(), eps))
        scale = ref.abs().mean().item()
        assert (y.double() - ref).abs().max().item() < 3e-6 * max(scale, 1.0)
        assert (y - y_ref).abs().max().item() < 3e-6 * max(scale, 1.0)
        mu = C.double().mean(1); var = C.double().var(1, unbiased=False)
        assert torch.allclose(mean.double(), mu, rtol=1e-5, atol=1e-6)
        assert torch.allclose(rstd.double(), 1 / torch.sqrt(var + eps), rtol=2e-6)
        # planes: one scale for the tensor, 22 bits of every element that is not tiny against the bound
        assert (out_p.inv == out_p.inv[0]).all() and out_p.inv[0] > 0
        bound = (gamma.abs().max() * N ** .5 + beta.abs().max()).item()
        assert bound / out_p.inv[0].item() < 65504 and bound / out_p.inv[0].item() >= 2 ** 14
        back = out_p.float().double()
        err = (back - y.double()).abs()
        assert (err <= 2.0 ** -21 * y.double().abs() + 2.0 ** -36 * bound).all(), err.max().item()
    # planes-only form (no fp32 y, no statistics wanted)
    out2 = planes.Planes(M, N, 'cuda')
    C2 = torch.empty(M, N, device='cuda')
    planes.gemm_ln(a0, b0, C2, bias, M, N, gamma, beta, eps, out2, 0, A1=a1, B1=b1)
    assert torch.equal(out2.t[:, :, :N], out_p.t[:, :, :N]) and torch.equal(C2, C)


def test_fused_layer_feeds_the_next_product_and_replays_in_a_graph():
    """two chained layers (the second reads the first one's planes), captured once and replayed: bit-identical to the eager launches"""
    from genrl_amd import planes
    M, N, K = 1024, 1024, 1024
    (a0, b0, _, _), bias, gamma, beta = _case(M, N, K, 0, seed=10)
    (_, b1, _, _), bias1, gamma1, beta1 = _case(M, N, N, 0, seed=20)
    bufs = dict(C0=torch.empty(M, N, device='cuda'), C1=torch.empty(M, N, device='cuda'), y1=torch.empty(M, N, device='cuda'))
    p0, p1 = planes.Planes(M, N, 'cuda'), planes.Planes(M, N, 'cuda')

    def run():
        planes.gemm_ln(a0, b0, bufs['C0'], bias, M, N, gamma, beta, 1e-3, p0, 0)
        planes.gemm_ln(p0, b1, bufs['C1'], bias1, M, N, gamma1, beta1, 1e-3, p1, 0, y=bufs['y1'])
    run(); torch.cuda.synchronize()
    want = bufs['y1'].clone()
    # layer 1's input is what layer 0's planes hold (22-bit operands)
    ref = F.silu(F.layer_norm(p0.float().double() @ b1.float().double().t() + bias1.double(), (N,), gamma1.double(), beta1.double(), 1e-3))
    assert (want.double() - ref).abs().max().item() < 1e-5
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            run()
    for _ in range(5):
        bufs['y1'].fill_(float('nan'))
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(bufs['y1'], want)
    planes.check_ln_failure()


def test_fused_form_is_refused_where_it_cannot_run():
    from genrl_amd import planes, streams
    from genrl_amd._lib import lib
    L = lib()
    assert L.genrl_gemm_h2_ln_ok(1024, 1024) == 1 and L.genrl_gemm_h2_ln_ok(1088, 1024) == 0       # 17 row blocks x 16 tiles > 256 CUs
    assert L.genrl_gemm_h2_ln_ok(3200, 512) == 0 and L.genrl_gemm_h2_ln_ok(2048, 512) == 1
    assert L.genrl_gemm_h2_ln_ok(1024, 1000) == 0 and L.genrl_gemm_h2_ln_ok(1024, 2048) == 0
    assert planes.gemm_ln_ok(1024, 1024)
    with streams.fork('detached'):
        assert not planes.gemm_ln_ok(1024, 1024)        # never on a side stream: two such launches must not run concurrently
    streams.join()


@pytest.mark.parametrize('BS,BL', [(32, 32), (8, 32), (16, 16)])
def test_imagination_update_with_fused_layers_equals_the_two_launch_form(BS, BL, monkeypatch):
    """update_imag_behavior at full width on 1024 / 256 rollout rows (c2 and its DP-4 per-rank size, c5's 256 rows): Dense -> LayerNorm -> SiLU as
    ONE launch (10 launches per rollout step) against product + LayerNorm launch (16): sampled latents identical up to near-ties (< 1e-4 of them), every metric within 2e-5,
    actor / critic gradients within 1e-4 of their norms (the fused form scales its operand planes by a bound instead of the row maximum:
    same 22 bits per element, other rounding)"""
    import detgen
    from param_shapes import agent_param_shapes
    from oracle import genrl_oracle as O
    from genrl_amd import config, noise as gnoise, planes
    from genrl_amd.agent import dreamer_utils as common
    from test_gpu_iteration import FakeClip
    monkeypatch.delenv('GENRL_PLANES_MIN_ROWS', raising=False)
    A, S, K, H, seed = 10, 32, 32, 15, 8
    ocfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H)
    p = detgen.det_state_dict(agent_param_shapes(ocfg), seed)
    gen = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, K, (BS, BL, S), generator=gen)
    post = dict(stoch=F.one_hot(idx, K).float(), deter=torch.tanh(torch.randn(BS, BL, 1024, generator=gen)), logit=torch.randn(BS, BL, S, K, generator=gen))
    nz = detgen.iteration_noise(BS, BL, S, K, A, H, seed=seed)['imag']

    def run(fused):
        monkeypatch.setattr(planes, 'LN_FUSED', fused)
        calls = []
        orig = planes.gemm_ln
        monkeypatch.setattr(planes, 'gemm_ln', lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        from genrl_amd import ops
        monkeypatch.setattr(ops, 'SEQ_C', False)           # (the Python twin of the C launch loop: its launches are countable here)
        zero = dict(lr=0.0, wd=0.0)
        cfg = config.default_cfg(BS, BL, device='cuda', imag_horizon=H, model_opt=zero, actor_opt=zero, critic_opt=zero)
        ag = config.make_agent(cfg, act_dim=A)
        ag.load_state_dict({k: v.cuda() for k, v in p.items()})
        ag.wm.viclip_model = FakeClip()
        grads = {}
        names = {id(q): n for n, q in ag.named_parameters()}
        common.Optimizer.grad_hook = lambda opt, params: grads.__setitem__(opt, {names[id(q)]: q.grad.detach().clone().cpu() for q in params})
        try:
            with gnoise.inject({'imag.act_eps': nz['act_eps'], 'imag.step_q': nz['step_q'], 'imag.target_init_q': nz['target_init_q']}):
                outputs = dict(post={k: v.cuda() for k, v in post.items()}, is_terminal=torch.zeros(BS, BL, device='cuda'))
                seq, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics={}, seq_data=None)
        finally:
            common.Optimizer.grad_hook = None
        torch.cuda.synchronize()
        planes.check_ln_failure()
        st = seq['stoch'].detach().argmax(-1).cpu() if isinstance(seq, dict) and 'stoch' in seq else None
        return {k: float(v) for k, v in mets.items()}, grads, len(calls), st
    m1, g1, n1, t1 = run(True)
    m0, g0, n0, t0 = run(False)
    assert n0 == 0 and n1 >= H * 6, (n1, n0)            # 4 policy layers + img_in + img_out per rollout step (+ the heads' layers)
    if t1 is not None:                                   # imagined latents: identical up to near-ties (another rounding of the same 22-bit operands)
        assert (t1 != t0).float().mean().item() < 1e-4
    for k, v in m0.items():
        if np.isfinite(v):
            np.testing.assert_allclose(m1[k], v, rtol=2e-5, atol=1e-6, err_msg=k)
    for ph in ('actor', 'critic'):
        num = np.sqrt(sum(float(((g1[ph][n].double() - g0[ph][n].double()) ** 2).sum()) for n in g0[ph]))
        den = np.sqrt(sum(float((g0[ph][n].double() ** 2).sum()) for n in g0[ph]))
        assert num <= 1e-4 * den, (ph, num, den)
