"""Round-5 kernels through the C-ABI, each against float64 / torch or against the kernel it specialises:
the 128 x 192 and two-workgroups-per-CU instantiations of the plane GEMM (csrc/gemm_planes.hip), the scan step kernels
(genrl_linear_sample32, genrl_onehot_gather_ln_fwd, genrl_onehot_fwd_masked / _bwd_masked, genrl_gru_gates_fwd_ld2 / _bwd_ldp:
EnsembleRSSM.obs_step / img_step, agent/dreamer_utils.py:432-473) and the planes-only LayerNorm forward."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from genrl_amd import planes, ops
    from genrl_amd._lib import lib, check
    return planes, ops, lib(), check


def _st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize('M,N,K', [(3200, 512, 512), (3200, 512, 1536), (3200, 1024, 512), (2048, 1024, 1024), (1100, 1536, 1024),
                                   (49152, 192, 320), (43700, 192, 64), (1024, 1024, 1024)])
def test_plane_gemm_round5_instantiations_vs_float64(env, M, N, K):
    """257+ tile launches with K <= 1536 take the two-stage ring with two workgroups per CU; N = 192 with >= 2048 64-tiles takes the
    128 x 192 tile (gemm_planes_hlw_kernel<false, 3>): same bound as every other plane product (1e-6 of the product's scale; the
    fp32 MFMAs give 3e-7 .. 5e-7 on the same data), padding columns untouched, bit-reproducible"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g)
    B = torch.randn(N, K, device='cuda', generator=g) * 0.1
    bias = torch.randn(N, device='cuda', generator=g)
    ldc = N + 4
    C = torch.full((M, ldc), float('nan'), device='cuda')
    Ap, Bp = planes.split(A), planes.split(B)
    planes.gemm(Ap, Bp, C, ldc, bias, M, N)
    ref = A.double() @ B.double().t() + bias.double()
    scale = (A.double().abs() @ B.double().abs().t()).mean().item()
    err = (C[:, :N].double() - ref).abs().max().item() / scale
    assert err < 1e-6, err
    assert torch.isnan(C[:, N:]).all()
    C2 = torch.full((M, ldc), float('nan'), device='cuda')
    planes.gemm(Ap, Bp, C2, ldc, bias, M, N)
    assert torch.equal(C[:, :N], C2[:, :N])


@pytest.mark.parametrize('M,S,U', [(8, 32, 512), (20, 32, 512), (64, 32, 1024), (70, 5, 48), (3, 2, 16)])
@pytest.mark.parametrize('masked', [False, True])
@pytest.mark.parametrize('mode', [False, True])
def test_linear_sample32(env, M, S, U, masked, mode):
    """head product + categorical sample in one launch: logits vs float64, sample / masked copy / class index identical to the
    one-hot kernel run on those logits with the same noise (mode: q = NULL -> argmax of the probabilities)"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(M + S + U)
    K = 32
    N = S * K
    x = torch.randn(M, U, device='cuda', generator=g)
    W = torch.randn(N, U, device='cuda', generator=g) * 0.3
    b = torch.randn(N, device='cuda', generator=g)
    q = None if mode else torch.empty(M * S, K, device='cuda').exponential_(1.0, generator=g)
    scale = (torch.rand(M, device='cuda', generator=g) > 0.4).float() if masked else None
    C = torch.full((M, N), float('nan'), device='cuda'); smp = torch.full((M, N), float('nan'), device='cuda')
    smp2 = torch.full((M, N), float('nan'), device='cuda'); idx = torch.full((M, S), -7, dtype=torch.int32, device='cuda')
    p = lambda t: t.data_ptr() if t is not None else None
    check(L.genrl_linear_sample32(p(x), U, p(W), U, p(b), p(C), N, p(q), p(smp), p(smp2), p(idx), p(scale), M, S, U, 0.99, _st()), 'ls32')
    ref = x.double() @ W.double().t() + b.double()
    sc = (x.double().abs() @ W.double().abs().t()).mean().item()
    assert ((C.double() - ref).abs().max().item() / sc) < 2e-6
    want = torch.empty(M, N, device='cuda')
    check(L.genrl_onehot_fwd(p(C), p(q), p(want), None, M * S, K, 0.99, _st()), 'onehot')
    assert torch.equal(smp, want)
    m = scale if masked else torch.ones(M, device='cuda')
    assert torch.equal(smp2, want * m[:, None])
    cls = want.reshape(M, S, K).argmax(-1).int()
    assert torch.equal(idx, torch.where(m[:, None] != 0, cls, torch.full_like(cls, -1)))


@pytest.mark.parametrize('M,S,K,N', [(8, 32, 32, 512), (64, 32, 32, 1024), (5, 4, 4, 32), (17, 8, 8, 48)])
def test_onehot_gather_ln(env, M, S, K, N):
    """one-hot latent x weight as a gather of rows of the transposed weight, fused with LayerNorm + SiLU: against the dense product
    + the LayerNorm kernel (float64 reference for the pre-activation); -1 entries contribute nothing"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(M + N)
    idx = torch.randint(0, K, (M, S), device='cuda', generator=g).int()
    idx[torch.rand(M, device='cuda', generator=g) < 0.3] = -1              # reset rows
    wT = torch.randn(S * K, N, device='cuda', generator=g) * 0.2
    xa = torch.randn(M, N, device='cuda', generator=g)
    gamma = 1 + 0.1 * torch.randn(N, device='cuda', generator=g); beta = 0.1 * torch.randn(N, device='cuda', generator=g)
    xpre = xa.clone()
    y = torch.full((M, N + 8), float('nan'), device='cuda'); mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
    check(L.genrl_onehot_gather_ln_fwd(idx.data_ptr(), S, K, wT.data_ptr(), N, xpre.data_ptr(), N, gamma.data_ptr(), beta.data_ptr(),
                                       y.data_ptr(), N + 8, mean.data_ptr(), rstd.data_ptr(), M, N, 1e-5, _st()), 'gather_ln')
    oh = F.one_hot(idx.clamp_min(0).long(), K).double() * (idx >= 0).double()[..., None]
    pre = xa.double() + oh.reshape(M, S * K) @ wT.double()
    assert torch.allclose(xpre.double(), pre, rtol=0, atol=2e-6 * pre.abs().max().item())
    ref = F.silu(F.layer_norm(pre, (N,), gamma.double(), beta.double(), 1e-5))
    assert torch.allclose(y[:, :N].double(), ref, rtol=0, atol=3e-6 * ref.abs().max().item())
    assert torch.isnan(y[:, N:]).all()
    assert torch.allclose(mean.double(), pre.mean(1), atol=1e-6) and torch.allclose(rstd.double(), 1 / (pre.var(1, unbiased=False) + 1e-5).sqrt(), rtol=1e-5)


@pytest.mark.parametrize('M,S,K', [(8, 32, 32), (3, 4, 4), (40, 8, 8)])
def test_onehot_masked_forms(env, M, S, K):
    """forward: the sample is the plain kernel's, the second output its row-scaled copy; backward: the straight-through gradient of
    upstream = g1 + scale * g2 equals the plain kernel's on that sum (g1 may be absent)"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(M + S)
    lg = torch.randn(M * S, K, device='cuda', generator=g); q = torch.empty(M * S, K, device='cuda').exponential_(1.0, generator=g)
    scale = (torch.rand(M, device='cuda', generator=g) > 0.5).float()
    s0 = torch.empty_like(lg); s1 = torch.empty_like(lg); s2 = torch.empty_like(lg); idx = torch.empty(M * S, dtype=torch.int32, device='cuda')
    check(L.genrl_onehot_fwd(lg.data_ptr(), q.data_ptr(), s0.data_ptr(), None, M * S, K, 0.99, _st()), 'fwd')
    check(L.genrl_onehot_fwd_masked(lg.data_ptr(), q.data_ptr(), s1.data_ptr(), s2.data_ptr(), idx.data_ptr(), scale.data_ptr(), S, M * S, K, 0.99,
                                    _st()), 'fwdm')
    rows = scale.repeat_interleave(S)
    assert torch.equal(s0, s1) and torch.equal(s2, s0 * rows[:, None])
    assert torch.equal(idx, torch.where(rows != 0, s0.argmax(-1).int(), torch.full((M * S,), -1, dtype=torch.int32, device='cuda')))
    g1 = torch.randn_like(lg); g2 = torch.randn_like(lg)
    for with_g1 in (True, False):
        up = (g1 if with_g1 else 0) + rows[:, None] * g2
        d0 = torch.empty_like(lg); d1 = torch.full_like(lg, 0.25)
        check(L.genrl_onehot_bwd(lg.data_ptr(), up.contiguous().data_ptr(), d0.data_ptr(), M * S, K, 0.99, 0, _st()), 'bwd')
        check(L.genrl_onehot_bwd_masked(lg.data_ptr(), g1.data_ptr() if with_g1 else None, g2.data_ptr(), scale.data_ptr(), S, d1.data_ptr(), M * S, K,
                                        0.99, 1, _st()), 'bwdm')
        assert torch.allclose(d1, d0 + 0.25, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('R,D,U', [(8, 512, 512), (40, 64, 32)])
def test_gate_block_on_column_blocks_of_wider_rows(env, R, D, U):
    """genrl_gru_gates_fwd_ld2 writes the scaled next state into the right half of a [x | h] row; genrl_gru_gates_bwd_ldp reads a slab
    that is a column block of a wider product's output: both bit-identical to the plain entries on compact copies"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(R + D)
    X = U + D
    pre = torch.randn(R, 3 * D, device='cuda', generator=g); xh = torch.randn(R, X, device='cuda', generator=g)
    gamma = 1 + 0.1 * torch.randn(3 * D, device='cuda', generator=g); beta = 0.1 * torch.randn(3 * D, device='cuda', generator=g)
    sc = (torch.rand(R, device='cuda', generator=g) > 0.5).float()
    h = xh[:, U:].contiguous()
    o0 = torch.empty(R, D, device='cuda'); o0b = torch.empty(R, D, device='cuda'); m0 = torch.empty(R, device='cuda'); r0 = torch.empty(R, device='cuda')
    check(L.genrl_gru_gates_fwd(pre.data_ptr(), h.data_ptr(), D, gamma.data_ptr(), beta.data_ptr(), o0.data_ptr(), D, o0b.data_ptr(), sc.data_ptr(),
                                m0.data_ptr(), r0.data_ptr(), R, D, 1e-5, _st()), 'gf')
    o1 = torch.empty(R, D, device='cuda'); nxt = torch.full((R, X), float('nan'), device='cuda'); m1 = torch.empty(R, device='cuda'); r1 = torch.empty(R, device='cuda')
    check(L.genrl_gru_gates_fwd_ld2(pre.data_ptr(), xh.data_ptr() + 4 * U, X, gamma.data_ptr(), beta.data_ptr(), o1.data_ptr(), D, nxt.data_ptr() + 4 * U, X,
                                    sc.data_ptr(), m1.data_ptr(), r1.data_ptr(), R, D, 1e-5, _st()), 'gf2')
    assert torch.equal(o0, o1) and torch.equal(nxt[:, U:], o0b) and torch.isnan(nxt[:, :U]).all() and torch.equal(m0, m1) and torch.equal(r0, r1)
    dout = torch.randn(R, D, device='cuda', generator=g); d2 = torch.randn(R, D, device='cuda', generator=g)
    wide = torch.randn(R, X, device='cuda', generator=g)
    slab = wide[:, U:].contiguous()
    ws = torch.empty(L.genrl_gru_ws_floats(R, D), device='cuda')
    outs = []
    for form in (0, 1):
        dpre = torch.empty(R, 3 * D, device='cuda'); dh = torch.empty(R, D, device='cuda'); dg = torch.empty(3 * D, device='cuda'); db = torch.empty(3 * D, device='cuda')
        if form == 0:
            check(L.genrl_gru_gates_bwd(dout.data_ptr(), D, d2.data_ptr(), sc.data_ptr(), pre.data_ptr(), h.data_ptr(), D, gamma.data_ptr(), beta.data_ptr(),
                                        m0.data_ptr(), r0.data_ptr(), dpre.data_ptr(), dh.data_ptr(), D, dg.data_ptr(), db.data_ptr(), ws.data_ptr(), R, D, 0,
                                        slab.data_ptr(), 1, 0, _st()), 'gb')
        else:
            check(L.genrl_gru_gates_bwd_ldp(dout.data_ptr(), D, d2.data_ptr(), sc.data_ptr(), pre.data_ptr(), xh.data_ptr() + 4 * U, X, gamma.data_ptr(),
                                            beta.data_ptr(), m0.data_ptr(), r0.data_ptr(), dpre.data_ptr(), dh.data_ptr(), D, dg.data_ptr(), db.data_ptr(),
                                            ws.data_ptr(), R, D, 0, wide.data_ptr() + 4 * U, 1, 0, X, _st()), 'gb2')
        outs.append((dpre, dh, dg, db))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_layernorm_forward_planes_only(env):
    """genrl_ln_act_fwd_h2 with y == NULL: the planes and statistics are those of the call that also writes the fp32 copy"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(4)
    M, N = 300, 1024
    x = torch.randn(M, N, device='cuda', generator=g); gamma = torch.rand(N, device='cuda', generator=g) + 0.5; beta = torch.randn(N, device='cuda', generator=g)
    res = []
    for with_y in (True, False):
        y = torch.empty(M, N, device='cuda') if with_y else None
        P = planes.Planes(M, N, 'cuda'); mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
        check(L.genrl_ln_act_fwd_h2(x.data_ptr(), N, gamma.data_ptr(), beta.data_ptr(), y.data_ptr() if with_y else None, N, mean.data_ptr(), rstd.data_ptr(),
                                    M, N, 1e-5, 1, P.ptr(0), P.ld, P.plane, P.inv_ptr(0), _st()), 'ln')
        res.append((P.t.clone(), P.inv.clone(), mean, rstd))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # a shape whose kernel cannot do without the fp32 copy refuses instead of writing through a null pointer
    xs = torch.randn(8, 32, device='cuda'); Ps = planes.Planes(8, 32, 'cuda'); gs = torch.ones(32, device='cuda')
    rc = L.genrl_ln_act_fwd_h2(xs.data_ptr(), 32, gs.data_ptr(), gs.data_ptr(), None, 32, res[0][2].data_ptr(), res[0][3].data_ptr(), 8, 32, 1e-5, 1,
                               Ps.ptr(0), Ps.ld, Ps.plane, Ps.inv_ptr(0), _st())
    assert rc == 1
