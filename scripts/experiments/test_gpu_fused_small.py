"""Few-row layers with the LayerNorm in the consumer's loader (csrc/fused_small.hip, ops.small_fused; the imagination rollout at
<= 256 rows -- the per-GPU size under data parallelism): the kernel against float64, the head kernel with the LayerNorm inside, and
the whole imagination update with the fused forward against the unfused one (same kernels otherwise) at full width."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def g(s):
    return torch.Generator().manual_seed(s)


def _partials(x):
    """reference partial statistics [N / 16][M][2] = (mean, M2) of 16 consecutive columns"""
    M, N = x.shape
    xb = x.double().reshape(M, N // 16, 16)
    mean = xb.mean(-1)
    m2 = ((xb - mean[..., None]) ** 2).sum(-1)
    return torch.stack([mean, m2], -1).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize('M,N,K0,K1,ln', [(128, 1024, 1024, 0, True), (128, 3072, 1024, 1024, True), (256, 1024, 1024, 12, False),
                                          (37, 32, 32, 16, True), (5, 48, 16, 0, False), (130, 1024, 1024, 1024, False), (16, 64, 64, 0, True)])
def test_small_fused_product(M, N, K0, K1, ln):
    from genrl_amd import ops
    a0 = torch.randn(M, K0, generator=g(1)) * 1.5 + 0.3
    w0 = torch.randn(N, K0, generator=g(2)) / K0 ** .5
    a1 = torch.randn(M, K1, generator=g(3)) if K1 else None
    w1 = torch.randn(N, K1, generator=g(4)) / K1 ** .5 if K1 else None
    b = 0.1 * torch.randn(N, generator=g(5))
    ga, be = 1 + 0.1 * torch.randn(K0, generator=g(6)), 0.1 * torch.randn(K0, generator=g(7))
    act = F.silu(F.layer_norm(a0.double(), (K0,), ga.double(), be.double(), 1e-3)) if ln else a0.double()
    ref = act @ w0.double().t() + b.double() + (a1.double() @ w1.double().t() if K1 else 0)
    d = lambda t: t.cuda().contiguous() if t is not None else None
    A0, W0, A1, W1, B, GA, BE = d(a0), d(w0), d(a1), d(w1), d(b), d(ga), d(be)
    st_in = d(_partials(a0).float()) if ln else None
    C = torch.full((M, N), float('nan'), device='cuda')
    st_out = torch.full((N // 16, M, 2), float('nan'), device='cuda')
    ops.small_fused(A0.data_ptr(), K0, W0.data_ptr(), K0, K0, C.data_ptr(), N, M, N,
                    ln=(st_in.data_ptr(), K0 // 16, GA, BE, 1e-3) if ln else None,
                    seg1=(A1.data_ptr(), K1, W1.data_ptr(), K1, K1) if K1 else None, bias=B, stats_out=st_out.data_ptr())
    err = (C.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6, err
    # the partial statistics it leaves describe ITS output
    sp = _partials(C.cpu())
    assert torch.allclose(st_out.cpu().double()[..., 0], sp[..., 0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(st_out.cpu().double()[..., 1], sp[..., 1], rtol=1e-4, atol=1e-6)
    # deterministic
    C2 = torch.empty_like(C)
    ops.small_fused(A0.data_ptr(), K0, W0.data_ptr(), K0, K0, C2.data_ptr(), N, M, N,
                    ln=(st_in.data_ptr(), K0 // 16, GA, BE, 1e-3) if ln else None,
                    seg1=(A1.data_ptr(), K1, W1.data_ptr(), K1, K1) if K1 else None, bias=B)
    assert torch.equal(C, C2)


@pytest.mark.parametrize('R,U,A', [(128, 1024, 10), (37, 32, 6), (256, 512, 9)])
def test_head_kernel_with_the_layernorm_inside(R, U, A):
    from genrl_amd import ops
    from genrl_amd._lib import lib, check
    pre = torch.randn(R, U, generator=g(1)) * 2 + 0.5
    W = torch.randn(2 * A, U, generator=g(2)) * 0.05; b = torch.randn(2 * A, generator=g(3)); eps = torch.randn(R, A, generator=g(4))
    ga, be = 1 + 0.1 * torch.randn(U, generator=g(5)), 0.1 * torch.randn(U, generator=g(6))
    y = F.silu(F.layer_norm(pre.double(), (U,), ga.double(), be.double(), 1e-3))
    ref = y @ W.double().t() + b.double()
    AP = (A + 3) // 4 * 4
    raw = torch.empty(R, 2 * A, device='cuda'); act = torch.zeros(R, AP, device='cuda')
    P, Wd, bd, ed, gd, bed = pre.cuda(), W.cuda(), b.cuda(), eps.cuda(), ga.cuda(), be.cuda()
    st = _partials(pre).float().cuda()
    check(lib().genrl_actor_head_ln_linear_fwd(P.data_ptr(), U, st.data_ptr(), U // 16, gd.data_ptr(), bed.data_ptr(), 1e-3, Wd.data_ptr(),
                                               bd.data_ptr(), ed.data_ptr(), raw.data_ptr(), act.data_ptr(), R, U, A, 0.1, 1.0, AP,
                                               torch.cuda.current_stream().cuda_stream), 'head_ln')
    assert ((raw.cpu().double() - ref).abs().max() / ref.abs().mean()).item() < 3e-6
    assert torch.allclose(act[:, :A], ops.actor_sample(raw, ed), rtol=0, atol=0)


@pytest.mark.parametrize('BS,BL', [(8, 16), (16, 16)])
def test_imagination_update_fused_forward_equals_unfused(BS, BL, monkeypatch):
    """update_imag_behavior on 128 / 256 start rows at full width (the fp32-operand rollout of the product's default policy):
    consumer-side LayerNorms (11 launches per step) against the row kernels (20): metrics within 1e-5, sampled latents identical,
    actor / critic gradients within 1e-4 of their norms"""
    import detgen
    from param_shapes import agent_param_shapes
    from oracle import genrl_oracle as O
    from genrl_amd import config, noise as gnoise, ops
    from genrl_amd.agent import dreamer_utils as common
    from test_gpu_iteration import FakeClip
    monkeypatch.setenv('GENRL_PLANES_MIN_ROWS', '320')        # the fp32-operand rollout (the product's policy below 192 rows) at both sizes
    A, S, K, H, seed = 10, 32, 32, 15, 8
    ocfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H)
    p = detgen.det_state_dict(agent_param_shapes(ocfg), seed)
    gen = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, K, (BS, BL, S), generator=gen)
    post = dict(stoch=F.one_hot(idx, K).float(), deter=torch.tanh(torch.randn(BS, BL, 1024, generator=gen)), logit=torch.randn(BS, BL, S, K, generator=gen))
    nz = detgen.iteration_noise(BS, BL, S, K, A, H, seed=seed)['imag']

    def run(fused):
        monkeypatch.setattr(ops, 'FUSED_SMALL', fused)
        seen = []
        orig = ops.small_fused
        monkeypatch.setattr(ops, 'small_fused', lambda *a, **k: (seen.append(1), orig(*a, **k))[1])
        cfg = config.default_cfg(BS, BL, device='cuda', imag_horizon=H, model_opt=dict(lr=0.0, wd=0.0), actor_opt=dict(lr=0.0, wd=0.0),
                                 critic_opt=dict(lr=0.0, wd=0.0))
        ag = config.make_agent(cfg, act_dim=A)
        ag.load_state_dict({k: v.cuda() for k, v in p.items()})
        ag.wm.viclip_model = FakeClip()
        grads = {}
        names = {id(q): n for n, q in ag.named_parameters()}
        common.Optimizer.grad_hook = lambda opt, params: grads.__setitem__(opt, {names[id(q)]: q.grad.detach().clone().cpu() for q in params})
        try:
            with gnoise.inject({'imag.act_eps': nz['act_eps'], 'imag.step_q': nz['step_q'], 'imag.target_init_q': nz['target_init_q']}):
                outputs = dict(post={k: v.cuda() for k, v in post.items()}, is_terminal=torch.zeros(BS, BL, device='cuda'))
                _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics={}, seq_data=None)
        finally:
            common.Optimizer.grad_hook = None
        monkeypatch.setattr(ops, 'small_fused', orig)
        return {k: float(v) for k, v in mets.items()}, grads, len(seen)
    m1, g1, n1 = run(True)
    m0, g0, n0 = run(False)
    assert n1 == H * 8 and n0 == 0, (n1, n0)        # 4 policy layers + img_in + GRU + img_out + dist = 8 fused products per step (+ head, gates, sample = 11 launches)
    for k, v in m0.items():
        if np.isfinite(v):
            np.testing.assert_allclose(m1[k], v, rtol=2e-5, atol=1e-6, err_msg=k)
    for ph in ('actor', 'critic'):
        num = np.sqrt(sum(float(((g1[ph][n].double() - g0[ph][n].double()) ** 2).sum()) for n in g0[ph]))
        den = np.sqrt(sum(float((g0[ph][n].double() ** 2).sum()) for n in g0[ph]))
        assert num <= 1e-4 * den, (ph, num, den)
