"""Pre-split GEMM operands (genrl_amd/csrc/gemm_planes.hip).  h2 planes (the product path: two fp16 planes of the row-scaled
value + the row's inverse scale): representation error bound, fp32-accurate products (vs float64), plane outputs of the row
kernels identical to the split of their fp32 outputs.  x3 planes (three bf16 planes, kept in the ABI): exact split, product vs
float64.  Through the C-ABI (ctypes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    from genrl_amd import planes, ops
    from genrl_amd._lib import lib, check
    return planes, ops, lib(), check


def _planes_equal_split(planes, P, y, row0=0):
    ref = planes.split(y.reshape(-1, y.shape[-1]).contiguous())
    R = ref.rows
    assert torch.equal(P.inv[row0:row0 + R], ref.inv)
    assert torch.equal(P.t[:, row0:row0 + R, :ref.cols], ref.t[:, :, :ref.cols])


def _repr_ok(back, x):
    """|a - (h + l / 2^11) / s| <= 2^-22 |a| + 2^-38 max_row|a| (fp16's subnormal floor under the row scale)"""
    tol = 2.0 ** -22 * x.double().abs() + 2.0 ** -38 * x.double().abs().amax(1, keepdim=True)
    return ((back.double() - x.double()).abs() <= tol).all()


def test_split_h2_representation_and_transposed(env):
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(0)
    # row magnitudes over 40 decades (the row scale absorbs them), 6 decades inside a row
    x = torch.randn(300, 200, device='cuda', generator=g) * torch.logspace(-20, 20, 300, device='cuda')[:, None] \
        * torch.logspace(-3, 3, 200, device='cuda')[None, :]
    x[0, 0] = 0.0; x[2, 2] = 16777216.0; x[3, 3] = -1.0; x[4] = 0.0
    p = planes.split(x)
    assert p.ld == 256 and _repr_ok(p.float(), x)
    assert (p.t[:, :, 200:] == 0).all()
    # the scale is a power of two that puts the row maximum into [2^14, 2^15); an all-zero row gets a finite scale
    amax = x.abs().amax(1)
    sc = amax.double() / p.inv.double()
    nz = amax > 0
    assert ((sc[nz] >= 2.0 ** 14) & (sc[nz] < 2.0 ** 15)).all()
    assert (torch.frexp(p.inv)[0] == 0.5).all() and torch.isfinite(p.inv).all() and (p.inv > 0).all()
    assert (p.float()[4] == 0).all()
    # typical accuracy for elements within 2^-12 of their row maximum: one fp32 rounding
    big = x.abs() >= x.abs().amax(1, keepdim=True) * 2.0 ** -12
    big &= x != 0
    rel = ((p.float().double() - x.double()).abs() / x.double().abs().clamp_min(1e-300))[big]
    assert rel.mean().item() < 2.0 ** -24 and rel.max().item() <= 2.0 ** -22
    # edges: tiny rows keep their relative accuracy (scaled up); an infinite element poisons its row (NaN products,
    # where an fp32 MFMA would carry the Inf through: DESIGN.md) and no other row
    tiny = torch.tensor([[1e-38, -3e-36, 7e-34, 1e-33], [1.0, 2.0, 3.0, 4.0]], device='cuda')
    assert _repr_ok(planes.split(tiny).float(), tiny)
    pinf = planes.split(torch.tensor([[float('inf'), 1.0], [1.0, 2.0]], device='cuda'))
    assert torch.equal(pinf.float()[1], torch.tensor([1.0, 2.0], device='cuda')) and not torch.isfinite(pinf.float()[0]).all()
    pt = planes.split(x, transpose=True)
    assert pt.rows == 200 and _repr_ok(pt.float(), x.t())
    # a column slice of a wider matrix (weight segments)
    ps = planes.split(x[:, 40:104])
    assert _repr_ok(ps.float(), x[:, 40:104])


def test_x3_variant_exact_split_and_product(env):
    """the three-bf16-plane format the kernel still offers (genrl_split_x3 / genrl_gemm_x3): exact split, fp32-accurate product"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(1)
    st = torch.cuda.current_stream().cuda_stream
    M, N, K = 200, 136, 192
    A = torch.randn(M, K, device='cuda', generator=g) * torch.logspace(-6, 6, K, device='cuda')
    B = torch.randn(N, K, device='cuda', generator=g) * 0.1
    def split3(x):
        out = torch.zeros(3, x.shape[0], planes.r64(x.shape[1]), dtype=torch.int16, device='cuda')
        check(L.genrl_split_x3(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), out.shape[2], out.shape[1] * out.shape[2], 0, st), 'split_x3')
        return out
    a3, b3 = split3(A), split3(B)
    f = lambda t: (t.to(torch.int32) << 16).view(torch.float32)
    assert torch.equal((f(a3[0]) + f(a3[1]) + f(a3[2]))[:, :K], A)
    C = torch.empty(M, N, device='cuda')
    for tile in (1, 2):
        prev = L.genrl_planes_force_tile(tile)
        try:
            check(L.genrl_gemm_x3(a3.data_ptr(), a3.shape[2], a3.shape[1] * a3.shape[2], b3.data_ptr(), b3.shape[2], b3.shape[1] * b3.shape[2],
                                  a3.shape[2], None, 0, 0, None, 0, 0, 0, C.data_ptr(), N, None, M, N, 0, st), 'gemm_x3')
        finally:
            L.genrl_planes_force_tile(prev)
        ref = A.double() @ B.double().t()
        assert ((C.double() - ref).abs().max() / (A.double().abs() @ B.double().abs().t()).mean()).item() < 1e-6


@pytest.mark.parametrize('M,N,K', [(64, 64, 64), (1024, 1024, 1024), (1000, 520, 192), (37, 10, 1024), (128, 3072, 2048),
                                   (4100, 256, 320), (16384, 1024, 256)])
@pytest.mark.parametrize('tile', [1, 2])
def test_gemm_h2_vs_float64(env, M, N, K, tile):
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    A = torch.randn(M, K, device='cuda', generator=g)
    B = torch.randn(N, K, device='cuda', generator=g) * 0.1
    bias = torch.randn(N, device='cuda', generator=g)
    ldc = (N + 3) // 4 * 4
    C = torch.full((M, ldc), float('nan'), device='cuda')
    prev = L.genrl_planes_force_tile(tile)
    try:
        planes.gemm(planes.split(A), planes.split(B), C, ldc, bias, M, N)
    finally:
        L.genrl_planes_force_tile(prev)
    ref = A.double() @ B.double().t() + bias.double()
    scale = (A.double().abs() @ B.double().abs().t()).mean().item()
    err = (C[:, :N].double() - ref).abs().max().item() / scale
    assert err < 1e-6, err                         # fp32-MFMA products on the same data: 3e-7 .. 5e-7
    if ldc > N:
        assert torch.isnan(C[:, N:]).all()         # padding columns untouched


def test_gemm_h2_segments_accumulate_offsets(env):
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(5)
    M, N, K0, K1 = 520, 384, 96, 40            # K0, K1 padded to 128 / 64 by the planes
    rows = 3 * M
    # (segment 1's operands live 2^20 / 2^-13 away from segment 0's: the accumulators are rescaled at the boundary)
    A0 = torch.randn(rows, K0, device='cuda', generator=g); A1 = torch.randn(rows, K1, device='cuda', generator=g) * 1e6
    W = torch.randn(N, K0 + K1, device='cuda', generator=g) * 0.2
    W[:, K0:] *= 1e-4
    C0 = torch.randn(2, M, N, device='cuda', generator=g)
    C = C0.clone()
    a0, a1 = planes.split(A0), planes.split(A1)
    planes.gemm(a0, planes.split(W[:, :K0]), C, N, None, M, N, accumulate=True, a_row0=M, A1=a1, B1=planes.split(W[:, K0:]), a1_row0=2 * M,
            c_off=M * N)
    ref = C0[1].double() + A0[M:2 * M].double() @ W[:, :K0].double().t() + A1[2 * M:].double() @ W[:, K0:].double().t()
    assert torch.equal(C[0], C0[0])
    assert ((C[1].double() - ref).abs().max() / ref.abs().mean()).item() < 2e-6
    # dgrad form: B = planes of W^T
    dy = torch.randn(M, N, device='cuda', generator=g)
    dx = torch.empty(M, K0 + K1, device='cuda')
    planes.gemm(planes.split(dy), planes.split(W, transpose=True), dx, K0 + K1, None, M, K0 + K1)
    ref = dy.double() @ W.double()
    assert ((dx.double() - ref).abs().max() / ref.abs().mean()).item() < 4e-6     # (fp32 MFMAs: 3e-6 .. 6e-6 on this measure)


def test_weight_cache_invalidation(env):
    planes, ops, L, check = env
    W = torch.nn.Parameter(torch.randn(70, 50, device='cuda'))
    p1 = planes.weight(W)
    assert planes.weight(W) is p1
    with torch.no_grad():
        W.mul_(2.0)
    assert _repr_ok(planes.weight(W).float(), W.detach() / 2)          # stale until told
    planes.invalidate()
    p2 = planes.weight(W)
    assert p2 is p1 and _repr_ok(p2.float(), W.detach())           # refreshed in place (graph-replay safe)
    assert _repr_ok(planes.weight(W, transpose=True, c0=8, c1=40).float(), W.detach()[:, 8:40].t())


@pytest.mark.parametrize('M,N', [(300, 1024), (70, 32), (64, 96), (33, 3072)])
def test_row_kernels_emit_planes(env, M, N):
    """LayerNorm(+SiLU) fwd / bwd with plane outputs: fp32 results bit-identical to the plain entry points, planes
    == split(fp32 output) (all kernel variants: block-per-row, lane-group, generic + split pass)"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(N)
    st = torch.cuda.current_stream().cuda_stream
    x = torch.randn(M, N, device='cuda', generator=g); dy = torch.randn(M, N, device='cuda', generator=g)
    gam = torch.randn(N, device='cuda', generator=g); bet = torch.randn(N, device='cuda', generator=g)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    P = planes.Planes(2 * M, N, 'cuda')
    check(L.genrl_ln_act_fwd(x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), y0.data_ptr(), N, mean.data_ptr(), rstd.data_ptr(),
                             M, N, 1e-5, 1, st), 'ln')
    check(L.genrl_ln_act_fwd_h2(x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), y1.data_ptr(), N, mean.data_ptr(),
                                rstd.data_ptr(), M, N, 1e-5, 1, P.ptr(M), P.ld, P.plane, P.inv_ptr(M), st), 'ln_h2')
    assert torch.equal(y0, y1)
    _planes_equal_split(planes, P, y1, row0=M)
    d0, d1 = torch.empty_like(x), torch.empty_like(x)
    check(L.genrl_ln_act_bwd(dy.data_ptr(), N, x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                             d0.data_ptr(), N, None, None, None, None, M, N, 1, 0, st), 'lnb')
    check(L.genrl_ln_act_bwd_h2(dy.data_ptr(), N, x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(),
                                rstd.data_ptr(), d1.data_ptr(), N, None, None, None, None, M, N, 1, 0, P.ptr(0), P.ld, P.plane,
                                P.inv_ptr(0), st), 'lnb_h2')
    assert torch.equal(d0, d1)
    _planes_equal_split(planes, P, d1, row0=0)
    # planes only (dx == NULL, the kernels that write planes themselves: 256 < N <= 4096): same planes, same parameter partials
    if 256 < N <= 4096:
        P2 = planes.Planes(M, N, 'cuda')
        ws = torch.empty(L.genrl_ln_ws_floats(M, N), device='cuda')
        ga, gb = torch.empty(2, N, device='cuda'), torch.empty(2, N, device='cuda')
        for k, (dxp, PP) in enumerate([(d1.data_ptr(), P), (None, P2)]):
            check(L.genrl_ln_act_bwd_h2(dy.data_ptr(), N, x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), dxp, N, ga[k].data_ptr(), gb[k].data_ptr(), None, ws.data_ptr(), M, N, 1, 0,
                                        PP.ptr(0), PP.ld, PP.plane, PP.inv_ptr(0), st), 'lnb_h2')
        assert torch.equal(P2.t[:, :M], P.t[:, :M]) and torch.equal(P2.inv[:M], P.inv[:M])
        assert torch.equal(ga[0], ga[1]) and torch.equal(gb[0], gb[1])
    else:       # the lane-group / generic variants split their fp32 output in a second pass: dx is required
        assert L.genrl_ln_act_bwd_h2(dy.data_ptr(), N, x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), None, N, None, None, None, None, M, N, 1, 0, P.ptr(0), P.ld, P.plane,
                                     P.inv_ptr(0), st) == 1


def test_gemm_h2_row_split_against_wave_quantisation(env):
    """17 x 1024 rows (the imagined trajectory incl. its start row) x 1024 columns = 1088 128x128 tiles = 4.25 rounds of the
    256 CUs: the last 1024 rows run as a second launch (64x64 tiles).  Same result as the unsplit product, also with two
    operand segments, a bias and accumulation."""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(17)
    M, N, K0, K1 = 17408, 1024, 128, 64
    A0 = torch.randn(M, K0, device='cuda', generator=g); A1 = torch.randn(M, K1, device='cuda', generator=g) * 30
    W = torch.randn(N, K0 + K1, device='cuda', generator=g) * 0.1
    bias = torch.randn(N, device='cuda', generator=g)
    C0 = torch.randn(M, N, device='cuda', generator=g)
    a0, a1, w0, w1 = planes.split(A0), planes.split(A1), planes.split(W[:, :K0]), planes.split(W[:, K0:])
    out = []
    for force in (0, 2):                          # 0: the default policy (split); 2: one launch of 128x128 tiles
        C = C0.clone()
        prev = L.genrl_planes_force_tile(force)
        try:
            planes.gemm(a0, w0, C, N, bias, M, N, accumulate=True, A1=a1, B1=w1)
        finally:
            L.genrl_planes_force_tile(prev)
        out.append(C)
    ref = C0.double() + bias.double() + A0.double() @ W[:, :K0].double().t() + A1.double() @ W[:, K0:].double().t()
    scale = (A0.double().abs() @ W[:, :K0].double().abs().t() + A1.double().abs() @ W[:, K0:].double().abs().t()).mean().item()
    for C in out:
        assert ((C.double() - ref).abs().max().item() / scale) < 1e-6
    assert torch.equal(out[0][:16384], out[1][:16384])          # the first launch is the same kernel on the same tiles


@pytest.mark.parametrize('R,D', [(50, 32), (130, 1024)])
def test_gru_onehot_actor_planes(env, R, D):
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(D)
    st = torch.cuda.current_stream().cuda_stream
    pre = torch.randn(R, 3 * D, device='cuda', generator=g); h = torch.randn(R, D, device='cuda', generator=g)
    gam = torch.randn(3 * D, device='cuda', generator=g); bet = torch.randn(3 * D, device='cuda', generator=g)
    out0, out1 = torch.empty_like(h), torch.empty_like(h)
    mean, rstd = torch.empty(R, device='cuda'), torch.empty(R, device='cuda')
    P = planes.Planes(R, D, 'cuda')
    check(L.genrl_gru_gates_fwd(pre.data_ptr(), h.data_ptr(), D, gam.data_ptr(), bet.data_ptr(), out0.data_ptr(), D, None, None,
                                mean.data_ptr(), rstd.data_ptr(), R, D, 1e-5, st), 'gru')
    check(L.genrl_gru_gates_fwd_h2(pre.data_ptr(), h.data_ptr(), D, gam.data_ptr(), bet.data_ptr(), out1.data_ptr(), D, None, None,
                                   mean.data_ptr(), rstd.data_ptr(), R, D, 1e-5, P.ptr(), P.ld, P.plane, P.inv_ptr(), st), 'gru_h2')
    assert torch.equal(out0, out1)
    _planes_equal_split(planes, P, out1)
    dout = torch.randn(R, D, device='cuda', generator=g)
    dp0, dp1 = torch.empty_like(pre), torch.empty_like(pre)
    dh0, dh1 = torch.empty_like(h), torch.empty_like(h)
    P3 = planes.Planes(R, 3 * D, 'cuda')
    check(L.genrl_gru_gates_bwd(dout.data_ptr(), D, None, None, pre.data_ptr(), h.data_ptr(), D, gam.data_ptr(), bet.data_ptr(),
                                mean.data_ptr(), rstd.data_ptr(), dp0.data_ptr(), dh0.data_ptr(), D, None, None, None, R, D, 0,
                                None, 0, 0, st), 'grub')
    check(L.genrl_gru_gates_bwd_h2(dout.data_ptr(), D, None, None, pre.data_ptr(), h.data_ptr(), D, gam.data_ptr(), bet.data_ptr(),
                                   mean.data_ptr(), rstd.data_ptr(), dp1.data_ptr(), dh1.data_ptr(), D, None, None, None, R, D, 0,
                                   None, 0, 0, P3.ptr(), P3.ld, P3.plane, P3.inv_ptr(), st), 'grub_h2')
    assert torch.equal(dp0, dp1) and torch.equal(dh0, dh1)
    _planes_equal_split(planes, P3, dp1)
    # one-hot sample / straight-through backward (S x K latents per row) and the actor head's action planes
    for S, K in ((4, 8), (8, 8), (32, 32)):      # plane rows of 32 (second-pass split), 64 and 1024 elements (one workgroup per row)
        lg = torch.randn(R, S * K, device='cuda', generator=g); q = torch.rand(R, S * K, device='cuda', generator=g) + 0.05
        s1 = torch.empty_like(lg)
        Ps = planes.Planes(R, S * K, 'cuda')
        check(L.genrl_onehot_fwd_h2(lg.data_ptr(), q.data_ptr(), s1.data_ptr(), None, R * S, K, 0.99, Ps.ptr(), S * K, Ps.ld, Ps.plane,
                                    Ps.inv_ptr(), st), 'oh_h2')
        assert torch.equal(s1, ops.onehot_sample(lg.reshape(R, S, K), q.reshape(R, S, K)).reshape(R, S * K))
        _planes_equal_split(planes, Ps, s1)
        gs = torch.randn(R, S * K, device='cuda', generator=g)
        d0, d1 = torch.empty_like(lg), torch.empty_like(lg)
        check(L.genrl_onehot_bwd(lg.data_ptr(), gs.data_ptr(), d0.data_ptr(), R * S, K, 0.99, 0, st), 'ohb')
        check(L.genrl_onehot_bwd_h2(lg.data_ptr(), gs.data_ptr(), d1.data_ptr(), R * S, K, 0.99, 0, Ps.ptr(), S * K, Ps.ld, Ps.plane,
                                    Ps.inv_ptr(), st), 'ohb_h2')
        assert torch.equal(d0, d1)
        _planes_equal_split(planes, Ps, d1)
    A = 10
    raw = torch.randn(R, 2 * A, device='cuda', generator=g); eps = torch.randn(R, A, device='cuda', generator=g)
    act = torch.zeros(R, 12, device='cuda')
    Pa = planes.Planes(R, A, 'cuda')
    check(L.genrl_actor_head_fwd_h2(raw.data_ptr(), eps.data_ptr(), act.data_ptr(), None, None, R, A, 0.1, 1.0, 12, Pa.ptr(), Pa.ld,
                                    Pa.plane, Pa.inv_ptr(), st), 'ah_h2')
    assert torch.equal(act[:, :A], ops.actor_sample(raw, eps))
    _planes_equal_split(planes, Pa, act[:, :A])
    assert (Pa.t[:, :, A:] == 0).all()


def test_mlp_chains_without_plane_operands_match_reference(monkeypatch):
    """GENRL_PLANES_MLP=0: the Dense+LayerNorm+SiLU chains back on the fp32-operand products (the default runs them on plane
    operands from 512 rows up, which every other test of the suite exercises) -- the tiny reference iteration must come
    out within the golden tolerances this way too."""
    import numpy as np
    from genrl_amd import config
    from test_gpu_iteration import run_product, check_vs_golden
    monkeypatch.setenv('GENRL_PLANES_MLP', '0')
    tiny_o = dict(deter=32, hidden=32, units=32, cnn_depth=4)
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product('tiny_iter.npz', True, config.tiny_overrides(), tiny_o)
    assert (outputs['post']['stoch'].argmax(-1).cpu().numpy() == g['post_idx']).all()
    check_vs_golden(g, mets_wm, mets, 2e-4)
    n = 0
    for key, val in g.items():
        if key.startswith('grad.'):
            _, ph, name = key.split('.', 2)
            np.testing.assert_allclose(grads[ph][name].numpy(), val, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(val).max()), err_msg=key)
            n += 1
    assert n > 50


def test_small_rollouts_take_the_fp32_operand_path(monkeypatch):
    """Default policy: below 192 rollout rows the imagination runs on the fp32-operand kernels (ops._Rollout /
    ops.ActorTape) -- and that path reproduces the reference's tiny iteration like the planes path does."""
    import numpy as np
    from genrl_amd import config, ops, ops_planes
    from test_gpu_iteration import run_product, check_vs_golden
    monkeypatch.delenv('GENRL_PLANES_MIN_ROWS', raising=False)
    assert ops_planes.min_rows() == 192
    seen = []
    orig = ops._Rollout.forward
    monkeypatch.setattr(ops._Rollout, 'forward', staticmethod(lambda *a, **k: (seen.append(1), orig(*a, **k))[1]))
    tiny_o = dict(deter=32, hidden=32, units=32, cnn_depth=4)
    g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = run_product('tiny_iter.npz', True, config.tiny_overrides(), tiny_o)
    assert seen, 'the fp32-operand rollout node did not run'
    check_vs_golden(g, mets_wm, mets, 2e-4)
    for key, val in g.items():
        if key.startswith('grad.actor.'):
            name = key.split('.', 2)[2]
            np.testing.assert_allclose(grads['actor'][name].numpy(), val, rtol=1e-3, atol=1e-5 * max(1.0, np.abs(val).max()), err_msg=key)


@pytest.mark.parametrize('R,U,A', [(130, 1024, 10), (37, 32, 6), (1024, 512, 9)])
def test_actor_head_linear_fused(env, R, U, A):
    """output layer + Normal head in one launch: raw == y W^T + b (vs float64), action == the head kernel on that raw,
    action planes == split(action)"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(R + U)
    st = torch.cuda.current_stream().cuda_stream
    y = torch.randn(R, U, device='cuda', generator=g); W = torch.randn(2 * A, U, device='cuda', generator=g) * 0.05
    b = torch.randn(2 * A, device='cuda', generator=g); eps = torch.randn(R, A, device='cuda', generator=g)
    AP = (A + 3) // 4 * 4
    raw = torch.empty(R, 2 * A, device='cuda'); act = torch.zeros(R, AP, device='cuda')
    P = planes.Planes(R, A, 'cuda')
    check(L.genrl_actor_head_linear_fwd(y.data_ptr(), U, W.data_ptr(), b.data_ptr(), eps.data_ptr(), raw.data_ptr(), act.data_ptr(), R, U, A,
                                        0.1, 1.0, AP, P.ptr(), P.ld, P.plane, P.inv_ptr(), st), 'head_linear')
    ref = y.double() @ W.double().t() + b.double()
    assert ((raw.double() - ref).abs().max() / ref.abs().mean()).item() < 2e-6
    assert torch.allclose(act[:, :A], ops.actor_sample(raw, eps), rtol=0, atol=0)
    _planes_equal_split(planes, P, act[:, :A])
    assert (act[:, A:] == 0).all()
    # without planes / without noise (mode)
    raw2 = torch.empty_like(raw); act2 = torch.zeros_like(act)
    check(L.genrl_actor_head_linear_fwd(y.data_ptr(), U, W.data_ptr(), b.data_ptr(), None, raw2.data_ptr(), act2.data_ptr(), R, U, A,
                                        0.1, 1.0, AP, None, 0, 0, None, st), 'head_linear')
    assert torch.equal(raw2, raw) and torch.equal(act2[:, :A], torch.tanh(raw[:, :A]))


@pytest.mark.parametrize('R,U,A,up', [(130, 1024, 10, True), (37, 32, 6, False)])
def test_actor_head_linear_bwd_fused(env, R, U, A, up):
    """d raw = head_bwd(dx W_a (+ upstream)): equals the unfused pair (product, genrl_actor_head_bwd)"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(R)
    st = torch.cuda.current_stream().cuda_stream
    dx = torch.randn(R, U, device='cuda', generator=g); WaT = torch.randn(A, U, device='cuda', generator=g) * 0.05
    raw = torch.randn(R, 2 * A, device='cuda', generator=g); eps = torch.randn(R, A, device='cuda', generator=g)
    AP = (A + 3) // 4 * 4
    dup = torch.zeros(R, AP, device='cuda'); dup[:, :A] = torch.randn(R, A, device='cuda', generator=g)
    draw = torch.empty(R, 2 * A, device='cuda')
    check(L.genrl_actor_head_linear_bwd(dx.data_ptr(), U, WaT.data_ptr(), dup.data_ptr() if up else None, AP, raw.data_ptr(), eps.data_ptr(),
                                        draw.data_ptr(), R, U, A, 0.1, 1.0, st), 'head_linear_bwd')
    dact = (dx.double() @ WaT.double().t() + (dup[:, :A].double() if up else 0)).float().contiguous()
    ref = torch.empty_like(draw)
    check(L.genrl_actor_head_bwd(dact.data_ptr(), raw.data_ptr(), eps.data_ptr(), ref.data_ptr(), R, A, 0.1, 1.0, A, st), 'head_bwd')
    assert torch.allclose(draw, ref, rtol=2e-5, atol=2e-6 * ref.abs().max().item())


@pytest.mark.parametrize('M,S', [(200, 4), (1024, 32)])
def test_gemm_with_sampling_epilogue(env, M, S):
    """genrl_gemm_h2_sample: logits identical to the plain plane product, sample / planes identical to the one-hot kernel
    run on those logits with the same noise"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(S)
    st = torch.cuda.current_stream().cuda_stream
    K, U = 32, 192
    N = S * K
    x = torch.randn(M, U, device='cuda', generator=g); W = torch.randn(N, U, device='cuda', generator=g) * 0.2
    b = torch.randn(N, device='cuda', generator=g); q = torch.empty(M, N, device='cuda').exponential_(1.0, generator=g)
    xp, wp = planes.split(x), planes.split(W)
    lg0 = torch.empty(M, N, device='cuda'); planes.gemm(xp, wp, lg0, N, b, M, N)
    lg1 = torch.full((M + 2, N), float('nan'), device='cuda'); smp = torch.full((M + 2, N), float('nan'), device='cuda')
    SP = planes.Planes(M + 2, N, 'cuda'); SP.t.fill_(-1); SP.inv.fill_(-1.0)
    planes.gemm_sample(xp, wp, lg1, N, b, M, N, q, N, 0.99, smp, N, SP, c_off=N, s_off=N, sp_row0=1)
    assert torch.equal(lg1[1:M + 1], lg0) and torch.isnan(lg1[0]).all() and torch.isnan(lg1[M + 1]).all()
    ref = ops.onehot_sample(lg0.reshape(M, S, K), q.reshape(M, S, K)).reshape(M, N)
    assert torch.equal(smp[1:M + 1], ref) and torch.isnan(smp[0]).all() and torch.isnan(smp[M + 1]).all()
    rp = planes.split(ref)
    assert torch.equal(SP.t[:, 1:M + 1], rp.t) and torch.equal(SP.inv[1:M + 1], rp.inv)
    assert (SP.t[:, 0] == -1).all() and (SP.t[:, M + 1] == -1).all() and SP.inv[0] == -1 and SP.inv[M + 1] == -1


def test_linear_on_plane_operands_matches_torch(env):
    """ops_planes.linear: forward, input / weight / bias gradients vs torch in float64, with and without caller-provided planes"""
    planes, ops, L, check = env
    from genrl_amd import ops_planes
    g = torch.Generator(device='cuda').manual_seed(3)
    M, K, N = 700, 192, 136
    x = torch.randn(M, K, device='cuda', generator=g, requires_grad=True)
    W = torch.nn.Parameter(torch.randn(N, K, device='cuda', generator=g) * 0.1); b = torch.nn.Parameter(torch.randn(N, device='cuda', generator=g))
    dy = torch.randn(M, N, device='cuda', generator=g)
    yr = x.double() @ W.double().t() + b.double()
    gx, gW, gb = torch.autograd.grad(yr, (x, W, b), dy.double())
    for given in (False, True):
        P = (planes.split(x.detach()), 0) if given else None
        planes.invalidate()
        y = ops_planes.linear(x, W, b, P)
        hx, hW, hb = torch.autograd.grad(y, (x, W, b), dy)
        for a, r in ((y, yr), (hx, gx), (hW, gW), (hb, gb)):
            assert ((a.double() - r).abs().max() / r.abs().mean()).item() < 5e-6


def test_noise_arena_slices_are_disjoint_and_replanned():
    """noise.new_step: one flat draw per kind, handed out as disjoint aligned slices in call order; a call pattern that differs
    from the plan falls back to its own draw and becomes the next plan"""
    from genrl_amd import noise
    dev = torch.device('cuda:0')
    shapes = [('exp', (3, 5, 7)), ('normal', (11,)), ('exp', (64, 64)), ('normal', (2, 3))]
    def step(shp):
        noise.new_step(dev)
        return [noise.draw(k, 'test.site', s, dev) for k, s in shp]
    first = step(shapes)                      # no plan yet: individual draws
    second = step(shapes)                     # planned: slices of the two flat tensors
    flat_e, flat_n = noise._flat['exp'][0], noise._flat['normal'][0]
    ptrs = []
    for (k, s), t in zip(shapes, second):
        assert tuple(t.shape) == s and t.is_contiguous() and t.data_ptr() % 256 == 0
        base = flat_e if k == 'exp' else flat_n
        assert base.data_ptr() <= t.data_ptr() < base.data_ptr() + 4 * base.numel()
        ptrs.append((t.data_ptr(), t.data_ptr() + 4 * t.numel()))
    ptrs.sort()
    assert all(a[1] <= b[0] for a, b in zip(ptrs, ptrs[1:]))
    assert (second[0] > 0).all() and abs(second[2].mean().item() - 1.0) < 0.1 and abs(second[2].var().item() - 1.0) < 0.2
    third = step(shapes[:2] + [('exp', (5,))])          # deviates at the third draw: own launch, same results semantics
    assert tuple(third[2].shape) == (5,) and (third[2] > 0).all()
    fourth = step(shapes[:2] + [('exp', (5,))])         # ... and is the plan now
    assert noise._flat['exp'][0].data_ptr() <= fourth[2].data_ptr() < noise._flat['exp'][0].data_ptr() + 4 * noise._flat['exp'][0].numel()


@pytest.mark.parametrize('M,NI,NJ', [(512, 128, 128), (1024, 1024, 1024), (16384, 1024, 1024), (2048, 1024, 2048), (1024, 3072, 1024),
                                     (4096, 200, 520), (640, 1024, 1536), (17408, 256, 1024), (64, 32, 16), (1088, 255, 1024),
                                     (192, 32, 48)])
@pytest.mark.parametrize('decades', [6, 20])
def test_gemm_h2_tn_weight_gradient_vs_float64(env, M, NI, NJ, decades):
    """genrl_gemm_h2_tn (csrc/gemm_planes_tn.hip): dW = dY^T X on the row-scaled planes of dY and X, the reduction over the ROW index
    -- transposing LDS reads, the row scales (inside the sum) folded into the fragments -- against float64, every output against
    ITS OWN sum of |terms|.  Row magnitudes spread over 2^+-6 per operand (what gradients and activations do) must stay within
    1e-6; over 2^+-20 per operand (products over 80 binary orders: rows more than 2^14 below the largest lose bits of their
    relative precision, rows more than 2^24 below vanish -- their whole contribution is smaller than the largest row's rounding)
    within 4e-6.  Odd and single stage counts, accumulate, ragged tiles, every split-K count, bit-reproducibility."""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(M + NI + NJ)
    rs = torch.exp2(torch.randint(-decades, decades + 1, (M, 1), device='cuda', generator=g).float())        # per-row magnitudes
    dY = torch.randn(M, NI, device='cuda', generator=g) * rs * 1e-3
    X = torch.randn(M, NJ, device='cuda', generator=g) / rs.flip(0) * 3.0
    dY[5] = 0.0; X[7] = 0.0                                        # all-zero rows (a finite scale, zero planes)
    pa, pb = planes.split(dY), planes.split(X)
    ldc = NJ + 4
    C0 = torch.randn(NI, ldc, device='cuda', generator=g)
    ref = dY.double().t() @ X.double()
    asum = dY.double().abs().t() @ X.double().abs()
    k = 1e-6 if decades <= 6 else 4e-6
    bound = k * asum + 0.02 * k * asum.mean()
    for acc in (False, True):
        C = C0.clone()
        planes.gemm_tn(pa, pb, C, ldc, NI, NJ, M, accumulate=acc)
        want = ref + (C0[:, :NJ].double() if acc else 0.0)
        err = (C[:, :NJ].double() - want).abs()
        tol = bound + (2e-7 * C0[:, :NJ].double().abs() if acc else 0.0)
        assert (err <= tol).all(), (acc, (err / tol).max().item())
        assert torch.equal(C[:, NJ:], C0[:, NJ:])                  # padding columns untouched
    # bit-reproducible (fixed split-K order)
    Cb = C0.clone(); planes.gemm_tn(pa, pb, Cb, ldc, NI, NJ, M, accumulate=True)
    assert torch.equal(Cb, C)


def test_gemm_h2_tn_row_blocks_of_a_rollout(env):
    """operands that are row ranges of larger plane sets (a rollout's time-major states: rows h * N + n) and an output that is a
    column block of a wider weight gradient"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(3)
    Mtot, M, NI, NJ, K = 1024 + 128, 1024, 256, 192, 512
    dY = torch.randn(Mtot, NI, device='cuda', generator=g)
    X = torch.randn(Mtot, NJ, device='cuda', generator=g)
    pa, pb = planes.split(dY), planes.split(X)
    dW = torch.zeros(NI, K, device='cuda')
    planes.gemm_tn(pa, pb, dW, K, NI, NJ, M, a_row0=64, b_row0=128, c_off=256)
    ref = dY[64:64 + M].double().t() @ X[128:128 + M].double()
    err = (dW[:, 256:256 + NJ].double() - ref).abs().max().item() / (dY.abs().double().t() @ X.abs().double()).mean().item()
    assert err < 2e-6, err
    assert (dW[:, :256] == 0).all() and (dW[:, 256 + NJ:] == 0).all()


def test_gemm_h2_tn_all_zero_gradient_gives_exact_zero(env):
    """a zero-initialised output layer (agent/dreamer.py:143-145,357-359) makes every gradient row of the trunk in front of it
    EXACTLY zero on the first step: the rows' inverse scales are 2^-123, their products with the activations' scales leave
    fp32's range -- the weight gradient must come out as 0, not NaN"""
    planes, ops, L, check = env
    g = torch.Generator(device='cuda').manual_seed(9)
    M, NI, NJ = 1024, 256, 512
    dY = torch.zeros(M, NI, device='cuda')
    X = torch.randn(M, NJ, device='cuda', generator=g)
    C0 = torch.randn(NI, NJ, device='cuda', generator=g)
    C = C0.clone()
    planes.gemm_tn(planes.split(dY), planes.split(X), C, NJ, NI, NJ, M, accumulate=True)
    assert torch.equal(C, C0)
    C = torch.full((NI, NJ), float('nan'), device='cuda')
    planes.gemm_tn(planes.split(dY), planes.split(X), C, NJ, NI, NJ, M)
    assert (C == 0).all()
    # half of the rows zero, the others not: the zero rows must not disturb the rest
    dY[::2] = torch.randn(M // 2, NI, device='cuda', generator=g) * 1e-4
    C = torch.empty(NI, NJ, device='cuda')
    planes.gemm_tn(planes.split(dY), planes.split(X), C, NJ, NI, NJ, M)
    ref = dY.double().t() @ X.double()
    asum = dY.double().abs().t() @ X.double().abs()
    assert ((C.double() - ref).abs() <= 1e-6 * asum + 1e-8 * asum.mean()).all()


@pytest.mark.parametrize('fp32_path', [False, True])
@pytest.mark.parametrize('tiny', [True, False])
def test_rollout_c_loop_is_the_python_loop(tiny, fp32_path, monkeypatch):
    """genrl_imagine_seq_fwd (csrc/seq.hip): the plane rollout's H-step launch loop from ONE C call -- the same 16 launches per step in the
    same order, so the imagination update's metrics and every actor / critic gradient are bit-identical to the per-launch Python loop
    (tiny widths: 4-class latents, the separate sampling kernel; full width: 32 classes, the sample in the product's epilogue).
    fp32_path: the fp32-operand rollout of the product's default policy below 192 rows (genrl_imagine_seq_f32_fwd / _bwd, 20 + 14 launches
    per step) instead of the plane rollout the suite forces on everywhere else"""
    if fp32_path:
        monkeypatch.delenv('GENRL_PLANES_MIN_ROWS', raising=False)
    import detgen
    from param_shapes import agent_param_shapes
    from oracle import genrl_oracle as O
    from genrl_amd import config, noise as gnoise, ops
    from genrl_amd.agent import dreamer_utils as common
    from test_gpu_iteration import FakeClip
    BS, BL, A, H, seed = 4, 16, 10, 15, 3
    S, K = (4, 4) if tiny else (32, 32)
    wid = dict(deter=32, hidden=32, units=32, cnn_depth=4) if tiny else {}
    ocfg = O.make_cfg(stoch=S, discrete=K, act_dim=A, horizon=H, **wid)
    p = detgen.det_state_dict(agent_param_shapes(ocfg), seed)
    gen = torch.Generator().manual_seed(seed)
    Dd = 32 if tiny else 1024
    idx = torch.randint(0, K, (BS, BL, S), generator=gen)
    post = dict(stoch=torch.nn.functional.one_hot(idx, K).float(), deter=torch.tanh(torch.randn(BS, BL, Dd, generator=gen)),
                logit=torch.randn(BS, BL, S, K, generator=gen))
    nz = detgen.iteration_noise(BS, BL, S, K, A, H, seed=seed)['imag']

    def run(flag):
        monkeypatch.setattr(ops, 'SEQ_C', flag)
        zero = dict(lr=0.0, wd=0.0)
        cfg = config.default_cfg(BS, BL, device='cuda', imag_horizon=H, model_opt=zero, actor_opt=zero, critic_opt=zero,
                                 **(config.tiny_overrides() if tiny else {}))
        ag = config.make_agent(cfg, act_dim=A)
        ag.load_state_dict({k: v.cuda() for k, v in p.items()})
        ag.wm.viclip_model = FakeClip()
        grads = {}
        names = {id(q): n for n, q in ag.named_parameters()}
        common.Optimizer.grad_hook = lambda opt, params: grads.__setitem__(opt, {names[id(q)]: q.grad.detach().clone() for q in params})
        try:
            with gnoise.inject({'imag.act_eps': nz['act_eps'], 'imag.step_q': nz['step_q'], 'imag.target_init_q': nz['target_init_q']}):
                outputs = dict(post={k: v.cuda() for k, v in post.items()}, is_terminal=torch.zeros(BS, BL, device='cuda'))
                _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics={}, seq_data=None)
        finally:
            common.Optimizer.grad_hook = None
        return {k: float(v) for k, v in mets.items()}, grads
    seen = []
    from genrl_amd._lib import lib
    m1, g1 = run(True)
    m0, g0 = run(False)
    assert m1 == m0, {k: (m1[k], m0[k]) for k in m0 if m1[k] != m0[k]}
    for ph in g0:
        for n in g0[ph]:
            assert torch.equal(g1[ph][n], g0[ph][n]), (ph, n)
