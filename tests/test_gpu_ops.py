"""GPU parity of every HIP op (through the C-ABI, genrl_amd/ops.py) against the CPU oracle on the
same seeded inputs: forward values and all input gradients.  fp32; tolerance stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import genrl_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs MI355X')
    import genrl_amd.ops as ops_
    return ops_


def g(seed):
    return torch.Generator().manual_seed(seed)


def compare(hip_fn, ref_fn, inputs, rtol=2e-5, atol=2e-5, grad_mask=None, nondiff=()):
    """inputs: list of CPU tensors.  Float tensors not in `nondiff` get gradients checked."""
    cpu = [t.clone().requires_grad_(t.is_floating_point() and i not in nondiff) for i, t in enumerate(inputs)]
    dev = [t.clone().cuda().requires_grad_(t.is_floating_point() and i not in nondiff) for i, t in enumerate(inputs)]
    r = ref_fn(*cpu); h = hip_fn(*dev)
    r = r if isinstance(r, (tuple, list)) else (r,)
    h = h if isinstance(h, (tuple, list)) else (h,)
    lr = lh = 0
    for i, (a, b) in enumerate(zip(r, h)):
        np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().numpy(), rtol=rtol, atol=atol, err_msg=f'out{i}')
        if a.requires_grad:
            w = torch.randn(a.shape, generator=g(100 + i))
            lr = lr + (a * w).sum(); lh = lh + (b * w.cuda()).sum()
    if torch.is_tensor(lr):
        lr.backward(); lh.backward()
        for i, (a, b) in enumerate(zip(cpu, dev)):
            if a.requires_grad and a.grad is not None:
                scale = max(1.0, a.grad.abs().max().item())
                np.testing.assert_allclose(b.grad.cpu().numpy(), a.grad.numpy(), rtol=rtol * 5, atol=atol * scale,
                                           err_msg=f'grad{i}')


@pytest.mark.parametrize('M,N,K', [(64, 64, 16), (1, 1, 1), (37, 53, 29), (130, 70, 1034), (32, 255, 96),
                                   (1024, 1024, 256), (2048, 4096, 64), (96, 16, 26), (9000, 48, 64), (640, 200, 2050)])
def test_linear(ops, M, N, K):
    x = torch.randn(M, K, generator=g(1)); W = torch.randn(N, K, generator=g(2)) / K ** 0.5
    b = torch.randn(N, generator=g(3))
    compare(ops.linear, F.linear, [x, W, b], rtol=1e-4, atol=1e-4)


def test_sgemm_exact_layout(ops):
    """transpose-detecting: asymmetric small-integer operands must be reproduced exactly"""
    M, N, K = 70, 45, 33
    A = torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 7 - 3
    B = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 3) % 5 - 2
    C = torch.empty(M, N, device='cuda')
    ops.sgemm(A.cuda(), K, 1, B.cuda(), K, 1, C, N, None, M, N, K)
    assert torch.equal(C.cpu(), A @ B.T)
    # row-contiguous operands + accumulate
    At, Bt = A.T.contiguous().cuda(), B.T.contiguous().cuda()
    ops.sgemm(At, 1, M, Bt, 1, N, C, N, None, M, N, K, accumulate=True)
    assert torch.equal(C.cpu(), 2 * (A @ B.T))


@pytest.mark.parametrize('M,N,K', [(1024, 255, 512), (1024, 512, 255), (255, 512, 1024), (300, 255, 255), (2048, 130, 1030),
                                   (16384, 255, 1024)])
def test_sgemm_padded_lines(ops, M, N, K):
    """Extents that are not a multiple of 4 in lines padded to one (the 255-bin two-hot heads in rows of 256): the
    vector-load kernels read the padding, which holds NaN here, and must not let it reach the result.  All four
    operand layouts, padded C, bias + accumulate; exact on small integers."""
    r4 = lambda n: (n + 3) // 4 * 4
    A = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 5 - 2)
    B = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 3) % 7 - 3)
    bias = torch.arange(N, dtype=torch.float32) % 3
    want = A.double() @ B.double().T

    def padded(t):                       # rows of t in lines of roundup4(cols), padding = NaN
        buf = torch.full((t.shape[0], r4(t.shape[1])), float('nan'))
        buf[:, :t.shape[1]] = t
        return buf.cuda()
    for (a, ars, aks) in ((padded(A), r4(K), 1), (padded(A.T), 1, r4(M))):
        for (b, brs, bks) in ((padded(B), r4(K), 1), (padded(B.T), 1, r4(N))):
            C = torch.full((M, r4(N)), 1.0, device='cuda')
            ops.sgemm(a, ars, aks, b, brs, bks, C, r4(N), bias.cuda(), M, N, K, accumulate=True)
            assert torch.equal(C[:, :N].cpu().double(), want + bias.double() + 1.0), (ars, aks, brs, bks)
            assert torch.equal(C[:, N:].cpu(), torch.ones(M, r4(N) - N))          # C's padding is not written


@pytest.mark.parametrize('M,N,K,lay', [(1024, 1024, 1024, 'kk'), (1024, 768, 2048, 'kr'), (4096, 1024, 512, 'rr'),
                                       (16384, 1024, 1024, 'kk'), (300, 200, 260, 'kk'), (96, 1728, 40000, 'rr')])
def test_sgemm_bf16_mode(ops, M, N, K, lay):
    """precision 16: the MFMA operands are rounded to bf16 (nearest even), products and sums stay fp32 -- so the
    result must equal the fp32 product of the bf16-rounded operands up to summation order, and differ from the
    fp32-mode result by bf16 rounding (~1e-2 relative of the operand scale)."""
    A = torch.randn(M, K, generator=g(1)); B = torch.randn(N, K, generator=g(2))
    a, ars, aks = (A.cuda(), K, 1) if lay[0] == 'k' else (A.T.contiguous().cuda(), 1, M)
    b, brs, bks = (B.cuda(), K, 1) if lay[1] == 'k' else (B.T.contiguous().cuda(), 1, N)
    C32 = torch.empty(M, N, device='cuda'); C16 = torch.empty(M, N, device='cuda'); C3 = torch.empty(M, N, device='cuda')
    prev = ops.set_gemm_precision('f32')
    try:
        ops.sgemm(a, ars, aks, b, brs, bks, C32, N, None, M, N, K)
        ops.set_gemm_precision('bf16')
        ops.sgemm(a, ars, aks, b, brs, bks, C16, N, None, M, N, K)
        ops.set_gemm_precision('bf16x3')         # fp32 through three exact bf16 terms per operand: fp32-sized error
        ops.sgemm(a, ars, aks, b, brs, bks, C3, N, None, M, N, K)
    finally:
        ops.set_gemm_precision(prev)
    exact = A.double() @ B.double().T
    e32 = (C32.cpu().double() - exact).abs().max().item(); e3 = (C3.cpu().double() - exact).abs().max().item()
    assert e3 <= 2.0 * e32 + 1e-6, (e3, e32)        # the split product is as accurate as the fp32 MFMAs
    want = (A.bfloat16().double() @ B.bfloat16().double().T)
    np.testing.assert_allclose(C16.cpu().double().numpy(), want.numpy(), rtol=1e-4, atol=2e-5 * K ** 0.5)
    err = (C16 - C32).abs().max().item()
    assert 1e-4 * K ** 0.5 < err < 3e-2 * K ** 0.5, err           # really a different (bf16) rounding, of the expected size


@pytest.mark.parametrize('M,N,K', [(20000, 108, 48), (16391, 48, 108), (17000, 48, 48), (16384, 3, 48), (30001, 100, 20),
                                   (20000, 300, 64), (16500, 1000, 96), (16384, 113, 92)])
def test_sgemm_tall_stream(ops, M, N, K):
    """M >= 16384 rows against a small B held in registers (sgemm_tall_kernel: the 3-channel ends of the image
    encoder / decoder): both B layouts, ragged M and N, bias + accumulate, exact on small integers; the same
    shapes must agree with the tiled kernels' answer on random data."""
    A = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 5 - 2)
    B = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 3) % 7 - 3)
    bias = torch.arange(N, dtype=torch.float32) % 3
    want = A.double() @ B.double().T
    ldc = (N + 3) // 4 * 4
    for (b, brs, bks) in ((B.cuda(), K, 1), (B.T.contiguous().cuda(), 1, N)):
        C = torch.full((M, ldc), 1.0, device='cuda')
        ops.sgemm(A.cuda(), K, 1, b, brs, bks, C, ldc, bias.cuda(), M, N, K, accumulate=True)
        assert torch.equal(C[:, :N].cpu().double(), want + bias.double() + 1.0), (brs, bks)
        assert torch.equal(C[:, N:].cpu(), torch.ones(M, ldc - N))
        C2 = torch.empty(M, ldc, device='cuda')
        ops.sgemm(A.cuda(), K, 1, b, brs, bks, C2, ldc, None, M, N, K)
        assert torch.equal(C2[:, :N].cpu().double(), want)
    Ar = torch.randn(M, K, generator=g(1)); Br = torch.randn(N, K, generator=g(2))
    C3 = torch.empty(M, ldc, device='cuda')
    ops.sgemm(Ar.cuda(), K, 1, Br.cuda(), K, 1, C3, ldc, None, M, N, K)
    np.testing.assert_allclose(C3[:, :N].cpu().numpy(), (Ar.double() @ Br.double().T).numpy(), rtol=1e-4, atol=1e-4 * K ** 0.5)


@pytest.mark.parametrize('M', [1, 4, 16, 17, 32])
@pytest.mark.parametrize('N,K', [(10, 8), (16, 520), (1000, 1034), (3072, 1024), (1024, 3073), (33, 5)])
def test_sgemm_skinny(ops, M, N, K):
    """M <= 32 with k-contiguous A takes the MFMA-16x16x4 weight-stream kernel: both B layouts,
    bias + accumulate, ragged K/N, exact on small integers and close on random data."""
    A = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 7 - 3)
    B = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 3) % 5 - 2)
    bias = torch.arange(N, dtype=torch.float32) % 3
    C = torch.full((M, N), 2.0, device='cuda')
    ops.sgemm(A.cuda(), K, 1, B.cuda(), K, 1, C, N, bias.cuda(), M, N, K, accumulate=True)      # B k-contiguous
    want = A.double() @ B.double().T + bias.double() + 2.0
    assert torch.equal(C.cpu().double(), want)
    Bt = B.T.contiguous().cuda()                                                                 # B row-contiguous
    C2 = torch.empty(M, N, device='cuda')
    ops.sgemm(A.cuda(), K, 1, Bt, 1, N, C2, N, None, M, N, K)
    assert torch.equal(C2.cpu().double(), A.double() @ B.double().T)
    Ar = torch.randn(M, K, generator=g(1)); Br = torch.randn(N, K, generator=g(2))
    C3 = torch.empty(M, N, device='cuda')
    ops.sgemm(Ar.cuda(), K, 1, Br.cuda(), K, 1, C3, N, None, M, N, K)
    np.testing.assert_allclose(C3.cpu().numpy(), (Ar.double() @ Br.double().T).numpy(), rtol=1e-4, atol=1e-4 * K ** 0.5)


@pytest.mark.parametrize('M,N,act', [(5, 32, 1), (300, 1024, 1), (64, 3072, 0), (33, 48, 1), (64, 48, 1), (1000, 96, 1),
                                     (4099, 192, 1), (777, 100, 0), (2050, 256, 1), (5000, 12, 1)])
def test_ln_act(ops, M, N, act):
    x = torch.randn(M, N, generator=g(1)) * 2 + 0.3
    ga = 1 + 0.1 * torch.randn(N, generator=g(2)); be = 0.1 * torch.randn(N, generator=g(3))
    ref = lambda x, ga, be: (F.silu if act else (lambda t: t))(F.layer_norm(x, (N,), ga, be, 1e-3))
    compare(lambda x, ga, be: ops.ln_act(x, ga, be, 1e-3, act), ref, [x, ga, be], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('M,K1,K2,N', [(9, 16, 10, 32), (300, 1024, 10, 1024), (64, 512, 0, 512)])
def test_dense_ln_act(ops, M, K1, K2, N):
    x1 = torch.randn(M, K1, generator=g(1)); x2 = torch.randn(M, K2, generator=g(2))
    W = torch.randn(N, K1 + K2, generator=g(3)) / (K1 + K2) ** .5; b = torch.randn(N, generator=g(4))
    ga = 1 + 0.1 * torch.randn(N, generator=g(5)); be = 0.1 * torch.randn(N, generator=g(6))
    if K2:
        ref = lambda x1, x2, W, b, ga, be: F.silu(F.layer_norm(F.linear(torch.cat([x1, x2], -1), W, b), (N,), ga, be, 1e-5))
        compare(lambda x1, x2, W, b, ga, be: ops.dense_ln_act(x1, x2, W, b, ga, be), ref, [x1, x2, W, b, ga, be], rtol=2e-4, atol=2e-5)
    else:
        ref = lambda x1, W, b, ga, be: F.silu(F.layer_norm(F.linear(x1, W, b), (N,), ga, be, 1e-5))
        compare(lambda x1, W, b, ga, be: ops.dense_ln_act(x1, None, W, b, ga, be), ref, [x1, W, b, ga, be], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('R,D', [(7, 12), (128, 1024)])
def test_gru_gates(ops, R, D):
    pre = torch.randn(R, 3 * D, generator=g(1)); h = torch.randn(R, D, generator=g(2))
    ga = 1 + 0.1 * torch.randn(3 * D, generator=g(3)); be = 0.1 * torch.randn(3 * D, generator=g(4))

    def ref(pre, h, ga, be):
        parts = F.layer_norm(pre, (3 * D,), ga, be, 1e-5)
        r, c, u = torch.chunk(parts, 3, -1)
        r = torch.sigmoid(r); c = torch.tanh(r * c); u = torch.sigmoid(u - 1.0)
        return u * c + (1 - u) * h
    compare(ops.gru_gates, ref, [pre, h, ga, be], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('T,B,I,D,masked', [(5, 3, 8, 12, True), (4, 32, 64, 1024, True), (6, 2, 16, 32, False)])
def test_gru_seq_and_step(ops, T, B, I, D, masked):
    x = torch.randn(T, B, I, generator=g(1)); h0 = torch.randn(B, D, generator=g(2))
    W = torch.randn(3 * D, I + D, generator=g(3)) / (I + D) ** .5
    ga = 1 + 0.1 * torch.randn(3 * D, generator=g(4)); be = 0.1 * torch.randn(3 * D, generator=g(5))
    mask = (torch.rand(T, B, generator=g(6)) > 0.3).float() if masked else None

    def cell(x, h, W, ga, be):
        parts = F.layer_norm(F.linear(torch.cat([x, h], -1), W), (3 * D,), ga, be, 1e-5)
        r, c, u = torch.chunk(parts, 3, -1)
        r = torch.sigmoid(r); c = torch.tanh(r * c); u = torch.sigmoid(u - 1.0)
        return u * c + (1 - u) * h

    def ref(x, h0, W, ga, be):
        h, outs = h0, []
        for t in range(T):
            if mask is not None:
                h = h * mask[t][:, None]
            h = cell(x[t], h, W, ga, be); outs.append(h)
        return torch.stack(outs, 0)
    compare(lambda x, h0, W, ga, be: ops.gru_seq(x, mask.cuda() if masked else None, h0, W, ga, be), ref,
            [x, h0, W, ga, be], rtol=2e-4, atol=2e-5)
    compare(ops.gru_step, cell, [x[0], h0, W, ga, be], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('T,B,I,D,masked', [(5, 3, 8, 12, True), (8, 32, 64, 1024, True), (6, 4, 16, 32, False), (4, 64, 64, 512, True)])
def test_gru_seq_c_loop_is_the_python_loop(ops, T, B, I, D, masked, monkeypatch):
    """genrl_gru_seq_fwd / _bwd (csrc/seq.hip): the scans' per-step launch loops run from ONE C call -- the same launches in the same
    order, so outputs and every gradient are bit-identical to the per-step launches from Python (B = 64: the tile + split-K products
    with a workspace; B <= 32: the weight-streaming kernel with K-split slabs in the backward)"""
    def run(flag):
        monkeypatch.setattr(ops, 'SEQ_C', flag)
        ts = [t.clone().cuda().requires_grad_(True) for t in (torch.randn(T, B, I, generator=g(1)), torch.randn(B, D, generator=g(2)),
                                                             torch.randn(3 * D, I + D, generator=g(3)) / (I + D) ** .5,
                                                             1 + 0.1 * torch.randn(3 * D, generator=g(4)), 0.1 * torch.randn(3 * D, generator=g(5)))]
        mask = (torch.rand(T, B, generator=g(6)) > 0.3).float().cuda() if masked else None
        y = ops.gru_seq(ts[0], mask, ts[1], ts[2], ts[3], ts[4])
        (y * torch.randn(y.shape, generator=g(7)).cuda()).sum().backward()
        return [y.detach()] + [t.grad for t in ts]
    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


@pytest.mark.parametrize('R,S,K', [(9, 4, 4), (64, 32, 32), (5, 3, 7)])
def test_onehot_and_kl(ops, R, S, K):
    lg = torch.randn(R, S, K, generator=g(1)) * 2; lq = torch.randn(R, S, K, generator=g(2)) * 2
    q = torch.empty(R * S, K).exponential_(1, generator=g(3))
    compare(lambda l: ops.onehot_sample(l, q.cuda()), lambda l: O.onehot_sample(l, q), [lg], rtol=1e-4, atol=1e-6)
    assert torch.equal(ops.onehot_mode(lg.cuda()).cpu(), O.onehot_mode(lg).detach())
    compare(ops.cat_kl, O.cat_kl, [lg, lq], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ops.cat_entropy(lg.cuda()).cpu().numpy(), O.cat_entropy(lg).numpy(), rtol=1e-5)


@pytest.mark.parametrize('forward,balance', [(False, 0.8), (True, 0.3)])
def test_kl_balance(ops, forward, balance):
    """The balanced free-nats KL loss as one node vs the oracle's elementwise restatement of EnsembleRSSM.kl_loss:
    loss, per-row value and both logits' gradients, with about half the rows under the free-nats floor."""
    R, S, K = 96, 32, 32
    post = torch.randn(R, S, K, generator=g(1)); prior = post + 0.3 * torch.randn(R, S, K, generator=g(2))
    kls = O.cat_kl(post, prior).sort().values
    free = float(0.5 * (kls[R // 2 - 1] + kls[R // 2]))          # between two rows: no row sits on the clamp's kink
    lhs, rhs = (prior, post) if forward else (post, prior)
    mix = balance if forward else 1 - balance
    hip = lambda a, b: ops.kl_balance(a, b, mix, free)
    ref = lambda a, b: O.kl_loss(b, a, forward, balance, free) if forward else O.kl_loss(a, b, forward, balance, free)
    compare(lambda a, b: hip(a, b)[0], lambda a, b: ref(a, b)[0], [lhs, rhs], rtol=1e-4, atol=1e-6)
    v = hip(lhs.cuda(), rhs.cuda())[1]
    assert not v.requires_grad
    np.testing.assert_allclose(v.cpu().numpy(), O.cat_kl(lhs, rhs).numpy(), rtol=1e-4, atol=1e-5)


def test_twohot(ops):
    lg = torch.randn(6, 5, 255, generator=g(1))
    x = torch.tensor([-30., -20., -3.3, 0., 0.5, 19.999]).reshape(6, 1, 1) * torch.tensor([1., .5, 1.7, -1., 1e-3]).reshape(1, 5, 1)
    compare(lambda l: ops.twohot_logprob(l, x.cuda()), lambda l: O.twohot_logprob(l, x), [lg], rtol=1e-4, atol=1e-5)
    compare(ops.twohot_mean, O.twohot_mean, [lg], rtol=1e-4, atol=1e-5)


def test_twohot_head_padded_rows(ops):
    """The 255-bin head end to end: linear hands out logits as a column slice of rows padded to 256, the two-hot
    kernels read them in place and return the gradient the same way, and linear's dgrad / wgrad / bias
    gradient consume that slice without a copy — against plain torch on the host."""
    B, T, K = 12, 32, 512
    x = torch.randn(B, T, K, generator=g(1)); W = torch.randn(255, K, generator=g(2)) / K ** 0.5
    b = torch.randn(255, generator=g(3)) * 0.1
    tgt = torch.randn(B, T, 1, generator=g(4)) * 3
    lg = ops.linear(x.cuda(), W.cuda(), b.cuda())
    assert lg.shape == (B, T, 255) and lg.stride() == (T * 256, 256, 1)
    hip = lambda x, W, b: (ops.twohot_logprob(ops.linear(x, W, b), tgt.cuda()), ops.twohot_mean(ops.linear(x, W, b)))
    ref = lambda x, W, b: (O.twohot_logprob(F.linear(x, W, b), tgt), O.twohot_mean(F.linear(x, W, b)))
    compare(hip, ref, [x, W, b], rtol=1e-4, atol=1e-4)


def test_lambda_return(ops):
    H, N = 16, 70
    r = torch.rand(H, N, 1, generator=g(1)); v = torch.randn(H + 1, N, 1, generator=g(2))
    ref = lambda r, v: O.lambda_return(r, v[:-1], 0.99 * torch.ones_like(r), v[-1], 0.95)
    compare(lambda r, v: ops.lambda_return(r, v, 0.99, 0.95), ref, [r, v], rtol=1e-5, atol=1e-6)


def test_mse_maxcos_actor(ops):
    m = torch.randn(3, 3, 16, 16, generator=g(1)) * 0.3
    o = torch.randint(0, 256, (3, 3, 16, 16), generator=g(2), dtype=torch.uint8)
    compare(lambda m: ops.mse_like(m, o.cuda()), lambda m: -((m - (o / 255.0 - 0.5)) ** 2).sum([1, 2, 3]), [m],
            rtol=1e-5, atol=1e-4)
    u = torch.randn(5, 7, 48, generator=g(3)); v = torch.randn(5, 7, 48, generator=g(4)) * torch.rand(5, 7, 1, generator=g(5)) * 2
    compare(lambda v: ops.maxcos(u.cuda(), v), lambda v: O.max_cosine_similarity(u, v), [v], rtol=1e-4, atol=1e-6)
    idx = torch.randint(0, 35, (5, 7), generator=g(6))
    compare(lambda v: ops.maxcos(u.cuda(), v, idx.cuda()), lambda v: O.max_cosine_similarity(u.reshape(35, 48)[idx], v),
            [v], rtol=1e-4, atol=1e-6)
    vt = v.clone(); vt[0] = u[0]      # exact ties |u| == |v|: torch.max splits the gradient
    compare(lambda v: ops.maxcos(u.cuda(), v), lambda v: O.max_cosine_similarity(u, v), [vt], rtol=1e-4, atol=1e-6)
    raw = torch.randn(9, 20, generator=g(7)); eps = torch.randn(9, 10, generator=g(8))

    def ref(raw):
        mean = torch.tanh(raw[:, :10]); std = 0.9 * torch.sigmoid(raw[:, 10:] + 2.0) + 0.1
        return mean + std * eps
    compare(lambda r: ops.actor_sample(r, eps.cuda()), ref, [raw], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('N,Hi,C,Co,k', [(2, 16, 3, 8, 4), (3, 31, 8, 16, 4), (2, 6, 16, 32, 4), (1, 63, 4, 4, 4),
                                         (8, 31, 48, 96, 4), (64, 14, 96, 192, 4), (2, 33, 12, 20, 5),
                                         (512, 31, 48, 96, 4),       # 96-wide rectangular GEMM tiles, both gather sides
                                         (1024, 14, 48, 256, 4)])    # 128x128 gather tiles (split-operand bf16 MFMAs by default)
def test_conv2d_s2(ops, N, Hi, C, Co, k):
    x = torch.randn(N, C, Hi, Hi, generator=g(1)); W = torch.randn(Co, C, k, k, generator=g(2)) / (C * k * k) ** .5
    b = torch.randn(Co, generator=g(3))
    ref = lambda x, W, b: F.conv2d(x, W, b, stride=2).permute(0, 2, 3, 1)
    compare(lambda x, W, b: ops.conv2d_s2(x.permute(0, 2, 3, 1).contiguous(), W, b), ref, [x, W, b], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('N,Hi,C,Co,k', [(3, 31, 8, 16, 4), (8, 14, 48, 96, 4)])
def test_conv_ln_act_fused(ops, N, Hi, C, Co, k):
    x = torch.randn(N, C, Hi, Hi, generator=g(1)); W = torch.randn(Co, C, k, k, generator=g(2)) / (C * k * k) ** .5
    b = torch.randn(Co, generator=g(3)); ga = 1 + 0.1 * torch.randn(Co, generator=g(4)); be = 0.1 * torch.randn(Co, generator=g(5))
    ref = lambda x, W, b, ga, be: F.silu(F.layer_norm(F.conv2d(x, W, b, stride=2).permute(0, 2, 3, 1), (Co,), ga, be, 1e-3))
    compare(lambda x, W, b, ga, be: ops.conv2d_s2(x.permute(0, 2, 3, 1).contiguous(), W, b, ln=(ga, be, 1e-3)), ref,
            [x, W, b, ga, be], rtol=2e-4, atol=2e-4)
    Wt = torch.randn(C, Co, k + 1, k + 1, generator=g(6)) / (C * k) ** .5
    reft = lambda x, W, b, ga, be: F.silu(F.layer_norm(F.conv_transpose2d(x, W, b, stride=2).permute(0, 2, 3, 1), (Co,), ga, be, 1e-3))
    compare(lambda x, W, b, ga, be: ops.convT2d_s2(x.permute(0, 2, 3, 1).contiguous(), W, b, ln=(ga, be, 1e-3)), reft,
            [x[:, :, :7, :7].contiguous(), Wt, b, ga, be], rtol=2e-4, atol=2e-4)


def test_conv2d_s2_u8(ops):
    o = torch.randint(0, 256, (2, 3, 64, 64), generator=g(1), dtype=torch.uint8)
    W = torch.randn(48, 3, 4, 4, generator=g(2)) / 7; b = torch.randn(48, generator=g(3))
    ref = lambda W, b: F.conv2d(o / 255.0 - 0.5, W, b, stride=2).permute(0, 2, 3, 1)
    compare(lambda W, b: ops.conv2d_s2(o.cuda(), W, b), ref, [W, b], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('N,Hi,Ci,Co,k', [(2, 1, 32, 8, 5), (2, 5, 8, 4, 5), (3, 13, 4, 6, 6), (1, 30, 6, 3, 6),
                                          (8, 13, 96, 48, 6), (16, 5, 192, 96, 5), (3, 7, 8, 12, 6), (128, 13, 96, 48, 6)])
def test_convT2d_s2(ops, N, Hi, Ci, Co, k):
    x = torch.randn(N, Ci, Hi, Hi, generator=g(1)); W = torch.randn(Ci, Co, k, k, generator=g(2)) / (Ci * k) ** .5
    b = torch.randn(Co, generator=g(3))
    ref = lambda x, W, b: F.conv_transpose2d(x, W, b, stride=2).permute(0, 2, 3, 1)
    compare(lambda x, W, b: ops.convT2d_s2(x.permute(0, 2, 3, 1).contiguous(), W, b), ref, [x, W, b], rtol=1e-4, atol=1e-4)
    if N <= 16:       # the NCHW-output form of the last decoder layer (overlap-add writes planes, dY patches read from planes)
        compare(lambda x, W, b: ops.convT2d_s2(x.permute(0, 2, 3, 1).contiguous(), W, b, out_nchw=True),
                lambda x, W, b: F.conv_transpose2d(x, W, b, stride=2), [x, W, b], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('M,N', [(5000, 3), (4194304, 3), (100000, 10), (70000, 16), (300, 3), (9000, 48), (100000, 255)])
def test_colsum(ops, M, N):
    """bias-gradient column sums: the wide kernel, the narrow-matrix kernel (3-channel image rows) and the accumulate
    form; exact on small integers (fixed summation order, sums below 2^24)."""
    x = ((torch.arange(M * N, dtype=torch.int64) * 7919) % 7 - 3).to(torch.float32).reshape(M, N)
    want = x.double().sum(0)
    out = ops.colsum(x.cuda())
    assert torch.equal(out.cpu().double(), want)
    acc = torch.full((N,), 2.0, device='cuda')
    ops.colsum(x.cuda(), out=acc, accumulate=True)
    assert torch.equal(acc.cpu().double(), want + 2.0)


def test_transpose_and_cat(ops):
    x = torch.randn(3, 4, 6, generator=g(1))
    compare(ops.transpose_last2, lambda x: x.transpose(1, 2).contiguous(), [x])
    a, b = torch.randn(5, 7, generator=g(2)), torch.randn(5, 3, generator=g(3))
    mask = torch.tensor([1., 0., 1., 1., 0.])
    out = ops.cat_cols([a.cuda(), b.cuda()], mask.cuda()).cpu()
    assert torch.equal(out, torch.cat([a, b], -1) * mask[:, None])


def test_adam_and_norm(ops):
    n = 100003
    p = torch.randn(n, generator=g(1)); gr = torch.randn(n, generator=g(2)) * 3
    pc = {'w': p.clone()}; st = {}
    pd, gd = p.cuda(), gr.cuda(); m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
    norm = torch.empty(1, device='cuda')
    for step in (1, 2):
        ref_norm = O.optimizer_step(pc, {'w': gr}, st, lr=1e-3, eps=1e-8, clip=100.0, wd=1e-6)
        ops.grad_norm(gd, norm)
        ops.adam_step(pd, gd, m, v, norm, 1.0, 100.0, 1e-3, 1e-8, 1e-6, step)
        np.testing.assert_allclose(norm.item(), ref_norm.item(), rtol=1e-5)
        np.testing.assert_allclose(pd.cpu().numpy(), pc['w'].numpy(), rtol=1e-5, atol=2e-6)


def test_align_index(ops):
    T, N, E, nf = 17, 6, 24, 8
    ct = torch.randn(T, N, E, generator=g(1)); ca = torch.randn(T, N, E, generator=g(2))
    scores = torch.stack([O.max_cosine_similarity(ct[:nf], ca[t:t + nf]).mean(0) for t in range(T - nf)], 0)
    best = scores.argmax(0)
    ref = (torch.clamp(torch.arange(T)[:, None] - best[None], min=0)) * N + torch.arange(N)[None]
    got = ops.align_index(ct.cuda(), ca.cuda(), nf).cpu()
    assert torch.equal(got, ref)


@pytest.mark.parametrize('M,N,K', [(1024, 1024, 10), (100, 300, 7), (257, 1000, 32), (1024, 20, 1024), (77, 10, 333),
                                   (2000, 32, 64)])
def test_sgemm_thin_products(ops, M, N, K):
    """Thin products (K <= 32: the action slice of the RSSM input layer; N <= 32: policy head, action gradient):
    all four operand layouts, bias + accumulate, exact on small integers."""
    A = (torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 5 - 2)
    B = ((torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 3) % 7 - 3)
    bias = torch.arange(N, dtype=torch.float32) % 3
    want = A.double() @ B.double().T
    C = torch.full((M, N), 1.0, device='cuda')
    ops.sgemm(A.cuda(), K, 1, B.cuda(), K, 1, C, N, bias.cuda(), M, N, K, accumulate=True)
    assert torch.equal(C.cpu().double(), want + bias.double() + 1.0)
    At, Bt = A.T.contiguous().cuda(), B.T.contiguous().cuda()
    for (a, ars, aks) in ((A.cuda(), K, 1), (At, 1, M)):
        for (b, brs, bks) in ((B.cuda(), K, 1), (Bt, 1, N)):
            C2 = torch.empty(M, N, device='cuda')
            ops.sgemm(a, ars, aks, b, brs, bks, C2, N, None, M, N, K)
            assert torch.equal(C2.cpu().double(), want), (ars, aks, brs, bks)


@pytest.mark.parametrize('M,N', [(1024, 1024), (300, 96), (4100, 512), (70, 3072), (4, 1024)])
def test_deferred_ln_parameter_reductions_match_the_immediate_ones(ops, M, N):
    """accumulate_params & 4: the LayerNorm backward leaves its per-workgroup partial rows, genrl_reduce_params_batch sums
    many sets in one launch -- bit-identical to the immediate reduction; two sets adding into the SAME buffers (a shared
    layer) are summed in successive launches by ops.defer_flush."""
    from genrl_amd._lib import lib, check
    L = lib()
    gen = torch.Generator(device='cuda').manual_seed(M + N)
    st = torch.cuda.current_stream().cuda_stream
    parts = L.genrl_ln_bwd_parts(M, N)
    assert parts > 0
    sets = []
    for k in range(3):
        x = torch.randn(M, N, device='cuda', generator=gen); dy = torch.randn(M, N, device='cuda', generator=gen)
        gam = torch.randn(N, device='cuda', generator=gen); bet = torch.randn(N, device='cuda', generator=gen)
        mean, rstd = x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
        sets.append((x, dy, gam, bet, mean.contiguous(), rstd.contiguous()))
    g0 = torch.randn(2, 3, N, device='cuda', generator=gen)          # [set 0 / sets 1+2 share][dgamma, dbeta, colsum]
    ref, got = g0.clone(), g0.clone()
    dx_ref = []
    for k, (x, dy, gam, bet, mean, rstd) in enumerate(sets):         # immediate, accumulating
        o = ref[min(k, 1)]
        dx = torch.empty_like(x); ws = torch.empty(L.genrl_ln_ws_floats(M, N), device='cuda')
        check(L.genrl_ln_act_bwd(dy.data_ptr(), N, x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                 dx.data_ptr(), N, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), ws.data_ptr(), M, N, 1, 1, st), 'ln_bwd')
        dx_ref.append(dx)
    ops.DEFER_REDUCTIONS, prev = True, ops.DEFER_REDUCTIONS
    try:
        ops.defer_begin()
        for k, (x, dy, gam, bet, mean, rstd) in enumerate(sets):
            o = got[min(k, 1)]
            dx = torch.empty_like(x); ws = torch.empty(L.genrl_ln_ws_floats(M, N), device='cuda')
            flag = ops.defer_reduce(M, N, ws, o[0], o[1], o[2])
            assert flag == 4
            check(L.genrl_ln_act_bwd(dy.data_ptr(), N, x.data_ptr(), N, gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                     dx.data_ptr(), N, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), ws.data_ptr(), M, N, 1, 1 | flag, st),
                  'ln_bwd')
            assert torch.equal(dx, dx_ref[k])
        assert torch.equal(got, g0)                                   # nothing added yet
        ops.defer_flush()
    finally:
        ops.DEFER_REDUCTIONS = prev
    torch.cuda.synchronize()
    assert torch.equal(got[0], ref[0])                               # one set: the same sum in the same order
    # two sets into one buffer: (g + s1) + s2 both ways
    assert torch.equal(got[1], ref[1])
    # shapes without per-workgroup partials refuse the flag
    assert L.genrl_ln_bwd_parts(8, 32) == 0
