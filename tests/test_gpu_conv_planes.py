"""The encoder / decoder convolutions with their products on h2 planes (genrl_amd/ops_conv_planes.py: patch-gathering plane GEMM,
transposing plane GEMM with gather for the weight gradients, uniform-scale planes from the channel-LayerNorm kernels) against
torch's fp32 conv2d / conv_transpose2d + LayerNorm + SiLU: outputs and every gradient, chains of layers (so that planes flow from
one layer's LayerNorm into the next layer's gather), the shapes of the 64 x 64 encoder / decoder at small batch, odd sizes that fall
back per product."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def g(s):
    return torch.Generator().manual_seed(s)


def _check(hip_fn, ref_fn, inputs, rtol, atol):
    cpu = [t.clone().double().requires_grad_(True) for t in inputs]
    dev = [t.clone().cuda().requires_grad_(True) for t in inputs]
    r, h = ref_fn(*cpu), hip_fn(*dev)
    np.testing.assert_allclose(h.detach().cpu().numpy(), r.detach().float().numpy(), rtol=rtol, atol=atol, err_msg='out')
    w = torch.randn(r.shape, generator=g(99))
    (r * w.double()).sum().backward(); (h * w.cuda()).sum().backward()
    for i, (a, b) in enumerate(zip(cpu, dev)):
        scale = max(1e-6, a.grad.abs().max().item())
        err = (b.grad.cpu().double() - a.grad).abs().max().item() / scale
        assert err <= 5e-5, (i, err)


def _ln_silu(y, ga, be):
    return F.silu(F.layer_norm(y, (y.shape[-1],), ga, be, 1e-3))


@pytest.mark.parametrize('N,Hi,C0,C1,C2', [(8, 31, 48, 96, 192), (64, 14, 96, 192, 384), (3, 31, 48, 96, 192), (16, 31, 56, 48, 64)])
@pytest.mark.parametrize('inner_fp32', [True, False])
def test_encoder_chain_on_planes(N, Hi, C0, C1, C2, inner_fp32):
    """two stride-2 k4 convolutions + channel-LN + SiLU: the second layer gathers its patches from the uniform planes the first
    layer's LayerNorm wrote; N = 3: pixel counts that are no multiple of 64 (weight gradients fall back to the fp32 kernels).
    inner_fp32 False: the first layer's LayerNorm writes the planes only, as inside Encoder._cnn (the N = 3 weight gradient then
    fills the fp32 values from the planes before it reads them: ops_conv_planes._need_fp32)"""
    from genrl_amd import ops, ops_conv_planes as cp, planes
    x = torch.randn(N, Hi, Hi, C0, generator=g(1))
    W1 = torch.randn(C1, C0, 4, 4, generator=g(2)) / (C0 * 16) ** .5; b1 = 0.1 * torch.randn(C1, generator=g(3))
    W2 = torch.randn(C2, C1, 4, 4, generator=g(4)) / (C1 * 16) ** .5; b2 = 0.1 * torch.randn(C2, generator=g(5))
    g1, e1 = 1 + 0.1 * torch.randn(C1, generator=g(6)), 0.1 * torch.randn(C1, generator=g(7))
    g2, e2 = 1 + 0.1 * torch.randn(C2, generator=g(8)), 0.1 * torch.randn(C2, generator=g(9))

    def ref(x, W1, b1, g1, e1, W2, b2, g2, e2):
        y = _ln_silu(F.conv2d(x.permute(0, 3, 1, 2), W1, b1, stride=2).permute(0, 2, 3, 1), g1, e1)
        return _ln_silu(F.conv2d(y.permute(0, 3, 1, 2), W2, b2, stride=2).permute(0, 2, 3, 1), g2, e2)

    def hip(x, W1, b1, g1, e1, W2, b2, g2, e2):
        xin = x * 1.0
        xin._planes = cp._uniform_split(xin.detach().reshape(-1, C0))            # (as a previous layer would have left them)
        y = cp.conv2d_s2(xin, W1, b1, (g1, e1, 1e-3), fp32_out=inner_fp32)
        assert y._planes is not None and (y._lazy is None) == inner_fp32
        out = cp.conv2d_s2(y, W2, b2, (g2, e2, 1e-3))
        hip.cell = y._lazy
        return out
    _check(hip, ref, [x, W1, b1, g1, e1, W2, b2, g2, e2], rtol=3e-4, atol=3e-4)
    if not inner_fp32:               # filled exactly when a product had to read it: the fp32 weight gradient of the ragged pixel count
        H2 = (((Hi - 4) // 2 + 1) - 4) // 2 + 1
        assert hip.cell[0] == ((N * H2 * H2) % 64 == 0), hip.cell


@pytest.mark.parametrize('N,Hi,C0,C1,C2,k1,k2', [(64, 1, 1536, 192, 96, 5, 5), (64, 5, 192, 96, 48, 5, 6), (5, 5, 192, 96, 48, 5, 6),
                                                 (128, 5, 64, 48, 56, 5, 6)])
@pytest.mark.parametrize('inner_fp32', [True, False])
def test_decoder_chain_on_planes(N, Hi, C0, C1, C2, k1, k2, inner_fp32):
    """two stride-2 transposed convolutions + channel-LN + SiLU: forward on planes of the input rows, the input gradient gathers
    patches of dY from uniform planes, the weight gradient sums x^T patches(dY) over the input pixels"""
    from genrl_amd import ops_conv_planes as cp
    x = torch.randn(N, Hi, Hi, C0, generator=g(1))
    W1 = torch.randn(C0, C1, k1, k1, generator=g(2)) / (C0 * k1) ** .5; b1 = 0.1 * torch.randn(C1, generator=g(3))
    W2 = torch.randn(C1, C2, k2, k2, generator=g(4)) / (C1 * k2) ** .5; b2 = 0.1 * torch.randn(C2, generator=g(5))
    g1, e1 = 1 + 0.1 * torch.randn(C1, generator=g(6)), 0.1 * torch.randn(C1, generator=g(7))
    g2, e2 = 1 + 0.1 * torch.randn(C2, generator=g(8)), 0.1 * torch.randn(C2, generator=g(9))

    def ref(x, W1, b1, g1, e1, W2, b2, g2, e2):
        y = _ln_silu(F.conv_transpose2d(x.permute(0, 3, 1, 2), W1, b1, stride=2).permute(0, 2, 3, 1), g1, e1)
        return _ln_silu(F.conv_transpose2d(y.permute(0, 3, 1, 2), W2, b2, stride=2).permute(0, 2, 3, 1), g2, e2)

    def hip(x, W1, b1, g1, e1, W2, b2, g2, e2):
        y = cp.convT2d_s2(x * 1.0, W1, b1, (g1, e1, 1e-3), fp32_out=inner_fp32)
        return cp.convT2d_s2(y, W2, b2, (g2, e2, 1e-3))
    _check(hip, ref, [x, W1, b1, g1, e1, W2, b2, g2, e2], rtol=3e-4, atol=3e-4)


@pytest.mark.parametrize('C1,subpixel', [(32, True), (96, True), (96, False)])
def test_fp32_output_follows_the_consumers_predicate(C1, subpixel, monkeypatch):
    """an inner layer skips its fp32 activation exactly when the NEXT layer's own predicates say it reads planes only --
    not by position.  cnn_depth = 32 (C1 = 32 < 48: the next convolution cannot gather from planes) and GENRL_SUBPIXEL=0 (the next
    transposed convolution with C < 512 input channels runs GEMM -> col2im on fp32 operands) keep the fp32 output; the default
    shapes skip it and never fill it.  Values and every gradient against torch either way."""
    from genrl_amd import ops_conv_planes as cp
    monkeypatch.setattr(cp, 'SUBPIXEL', subpixel)
    monkeypatch.setattr(cp, 'KR_MIN_K', 512)             # the product's default (the suite's environment lowers it to 0: GEMM -> col2im on planes at every width)
    N, Hi, C0, C2 = 64, 31, 48, 64
    x = torch.randn(N, Hi, Hi, C0, generator=g(1))
    W1 = torch.randn(C1, C0, 4, 4, generator=g(2)) / (C0 * 16) ** .5; b1 = 0.1 * torch.randn(C1, generator=g(3))
    W2 = torch.randn(C2, C1, 4, 4, generator=g(4)) / (C1 * 16) ** .5; b2 = 0.1 * torch.randn(C2, generator=g(5))
    g1, e1 = 1 + 0.1 * torch.randn(C1, generator=g(6)), 0.1 * torch.randn(C1, generator=g(7))
    g2, e2 = 1 + 0.1 * torch.randn(C2, generator=g(8)), 0.1 * torch.randn(C2, generator=g(9))
    cells = {}

    def ref(x, W1, b1, g1, e1, W2, b2, g2, e2):
        y = _ln_silu(F.conv2d(x.permute(0, 3, 1, 2), W1, b1, stride=2).permute(0, 2, 3, 1), g1, e1)
        return _ln_silu(F.conv2d(y.permute(0, 3, 1, 2), W2, b2, stride=2).permute(0, 2, 3, 1), g2, e2)

    def hip(x, W1, b1, g1, e1, W2, b2, g2, e2):
        xin = x * 1.0
        xin._planes = cp._uniform_split(xin.detach().reshape(-1, C0))
        y = cp.conv2d_s2(xin, W1, b1, (g1, e1, 1e-3), fp32_out=('conv', 4))
        cells['enc'] = y._lazy
        return cp.conv2d_s2(y, W2, b2, (g2, e2, 1e-3))
    _check(hip, ref, [x, W1, b1, g1, e1, W2, b2, g2, e2], rtol=3e-4, atol=3e-4)
    if C1 < 48:
        assert cells['enc'] is None                      # the consumer reads fp32: written by the LayerNorm, no rebuild
    else:
        assert cells['enc'] == [True]                    # skipped and never needed (N H2 H2 = 64 * 36 rows: a multiple of 64)

    # decoder: 5 x 5 x 192 -> 13 x 13 x 96 -> 30 x 30 x 48; the second layer takes the sub-pixel form unless it is switched off
    Nd, Hd, D0, D1, D2, k1, k2 = 64, 5, 192, 96, 48, 5, 6
    xd = torch.randn(Nd, Hd, Hd, D0, generator=g(11))
    V1 = torch.randn(D0, D1, k1, k1, generator=g(12)) / (D0 * k1) ** .5; c1 = 0.1 * torch.randn(D1, generator=g(13))
    V2 = torch.randn(D1, D2, k2, k2, generator=g(14)) / (D1 * k2) ** .5; c2 = 0.1 * torch.randn(D2, generator=g(15))
    h1, f1 = 1 + 0.1 * torch.randn(D1, generator=g(16)), 0.1 * torch.randn(D1, generator=g(17))
    h2, f2 = 1 + 0.1 * torch.randn(D2, generator=g(18)), 0.1 * torch.randn(D2, generator=g(19))

    def dref(x, W1, b1, g1, e1, W2, b2, g2, e2):
        y = _ln_silu(F.conv_transpose2d(x.permute(0, 3, 1, 2), W1, b1, stride=2).permute(0, 2, 3, 1), g1, e1)
        return _ln_silu(F.conv_transpose2d(y.permute(0, 3, 1, 2), W2, b2, stride=2).permute(0, 2, 3, 1), g2, e2)

    def dhip(x, W1, b1, g1, e1, W2, b2, g2, e2):
        y = cp.convT2d_s2(x * 1.0, W1, b1, (g1, e1, 1e-3), fp32_out=('convT', D2, k2))
        cells['dec'] = y._lazy
        return cp.convT2d_s2(y, W2, b2, (g2, e2, 1e-3))
    _check(dhip, dref, [xd, V1, c1, h1, f1, V2, c2, h2, f2], rtol=3e-4, atol=3e-4)
    if subpixel:
        assert cells['dec'] == [True]                    # sub-pixel gather forward, gathered weight gradient: fp32 never read
    else:
        assert cells['dec'] is None                      # GEMM -> col2im on fp32 operands: the LayerNorm wrote them


def test_uniform_planes_of_the_channel_layernorm():
    """genrl_ln_act_fwd_h2u: one scale for the whole tensor, from the parameters alone; the planes reproduce the fp32 output to
    2^-22 of the scale's range; genrl_split_h2u: the exact-maximum variant"""
    from genrl_amd import ops_conv_planes as cp, planes
    pre = torch.randn(5000, 96, generator=g(1)).cuda() * 3
    ga = (1 + 0.2 * torch.randn(96, generator=g(2))).cuda(); be = (0.3 * torch.randn(96, generator=g(3))).cuda()
    y, mean, rstd, P, lazy = cp._ln_fwd(pre, ga, be, 1e-3, True)
    assert lazy is None
    y2, mean2, rstd2, P2, lazy2 = cp._ln_fwd(pre, ga, be, 1e-3, True, want_fp32=False)     # planes only: the same planes, y filled on demand
    assert lazy2 == [True] and torch.equal(P2.t, P.t) and torch.equal(P2.inv, P.inv) and torch.equal(mean2, mean) and torch.equal(rstd2, rstd)
    cp._need_fp32(y2, lazy2, P2)
    assert lazy2 == [False] and torch.equal(y2, P.float())
    ref = F.silu(F.layer_norm(pre, (96,), ga, be, 1e-3))
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    assert (P.inv == P.inv[0]).all() and P.ld == 128 and (P.t[:, :, 96:] == 0).all()
    bound = ga.abs().max() * 96 ** 0.5 + be.abs().max()
    assert float(P.inv[0]) * 2 ** 14 <= float(bound) < float(P.inv[0]) * 2 ** 15
    assert ((P.float() - y).abs() <= 2.0 ** -21 * float(bound)).all() and ((P.float() - y).abs() <= 2.0 ** -22 * y.abs() + 2.0 ** -37 * float(bound)).all()
    x = torch.randn(777, 48, generator=g(4)).cuda() * 1e-5
    U = cp._uniform_split(x)
    assert (U.inv == U.inv[0]).all() and float(x.abs().max()) / float(U.inv[0]) < 2 ** 15 and float(x.abs().max()) / float(U.inv[0]) >= 2 ** 14
    assert ((U.float() - x).abs() <= 2.0 ** -22 * x.abs() + 2.0 ** -37 * float(x.abs().max())).all()


@pytest.mark.parametrize('N,Hi,Wi,Ci,Co,k', [(4, 13, 13, 96, 48, 6), (2, 5, 7, 192, 96, 6), (3, 5, 5, 192, 96, 5), (64, 13, 13, 96, 48, 6),
                                             (2, 6, 6, 48, 4, 4), (1, 3, 2, 64, 12, 2)])
def test_subpixel_gather_form_of_the_transposed_convolution(N, Hi, Wi, Ci, Co, k):
    """genrl_gemm_h2_subpixel (+ genrl_pad_planes, genrl_subpixel_weight): ConvTranspose2d(k, stride 2) forward as ONE product over
    the T x T patches of the zero-padded input with the pixel-shuffle epilogue, against torch in float64 -- even kernels, an odd one
    (k = 5 as 6 with a zero tap: the last output row / column is dropped), non-square images, channel counts off the tile sizes"""
    from genrl_amd import ops_conv_planes as cp
    x = torch.randn(N, Hi, Wi, Ci, generator=g(1))
    W = torch.randn(Ci, Co, k, k, generator=g(2)) / (Ci * k) ** .5
    b = 0.1 * torch.randn(Co, generator=g(3))
    ref = F.conv_transpose2d(x.double().permute(0, 3, 1, 2), W.double(), b.double(), stride=2).permute(0, 2, 3, 1)
    xd = x.cuda()
    xp = cp._uniform_split(xd.reshape(-1, Ci))
    Wp = W.permute(0, 2, 3, 1).contiguous().cuda()                     # (ci, kh, kw, co)
    Ho, Wo = 2 * (Hi - 1) + k, 2 * (Wi - 1) + k
    out = torch.full((N, Ho, Wo, Co), float('nan'), device='cuda')
    cp._subpixel(xp, N, Hi, Wi, Ci, Co, k, Wp, k * k * Co, 1, Co, b.cuda(), out)
    err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    # the original (Ci, Co, k, k) layout through its own strides, no bias
    out2 = torch.empty_like(out)
    cp._subpixel(xp, N, Hi, Wi, Ci, Co, k, W.cuda(), Co * k * k, k * k, 1, None, out2)
    assert torch.equal(out2, out - b.cuda())  or (out2 - (out - b.cuda())).abs().max().item() < 1e-5


@pytest.mark.parametrize('N,Hi,Ci,Co,k', [(4, 31, 48, 96, 4), (8, 14, 96, 192, 4), (2, 6, 192, 384, 4), (2, 30, 48, 96, 4)])
def test_subpixel_gather_form_of_the_convolution_input_gradient(N, Hi, Ci, Co, k):
    """the same product with the channel roles swapped = the input gradient of Conv2d(k, stride 2): against autograd in float64;
    Hi = 31: the last input row / column is reached by no patch and must come back as zeros"""
    from genrl_amd import ops_conv_planes as cp
    x = torch.randn(N, Ci, Hi, Hi, generator=g(1)).double().requires_grad_(True)
    W = torch.randn(Co, Ci, k, k, generator=g(2)) / (Ci * k * k) ** .5
    y = F.conv2d(x, W.double(), None, stride=2)
    dy = torch.randn(y.shape, generator=g(3))
    y.backward(dy.double())
    ref = x.grad.permute(0, 2, 3, 1)
    Ho = y.shape[2]
    dyp = cp._uniform_split(dy.permute(0, 2, 3, 1).contiguous().cuda().reshape(-1, Co))
    Wp = W.permute(0, 2, 3, 1).contiguous().cuda()                     # (co, kh, kw, ci)
    T = k // 2
    full = Hi <= 2 * (Ho + T - 1)
    dx = (torch.empty if full else torch.zeros)(N, Hi, Hi, Ci, device='cuda')
    cp._subpixel(dyp, N, Ho, Ho, Co, Ci, k, Wp, k * k * Ci, 1, Ci, None, dx)
    err = (dx.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err


@pytest.mark.parametrize('N,Hi,Wi,Co,nchw', [(4, 30, 30, 3, True), (3, 13, 9, 3, True), (2, 62, 62, 3, True), (5, 30, 30, 3, False), (2, 7, 30, 4, True),
                                             (1, 1, 1, 1, False)])
@pytest.mark.parametrize('bwd', [False, True])
def test_direct_three_channel_transposed_convolution(N, Hi, Wi, Co, nchw, bwd, monkeypatch):
    """genrl_convt_small_co_fwd: the decoder's last layer (48 -> 3 channels, k 6, stride 2) in gather form on the fp32 matrix cores,
    NCHW frames out -- against torch's conv_transpose2d in float64 (forward) and with every gradient through the unchanged backward;
    widths that are no multiple of the 16-position blocks, the 128 px decoder's 62 x 62 input, NHWC output, 4 and 1 channels"""
    from genrl_amd import ops
    Ci, k = 48, 6
    x = torch.randn(N, Hi, Wi, Ci, generator=g(1))
    W = torch.randn(Ci, Co, k, k, generator=g(2)) / (Ci * k) ** .5
    b = 0.1 * torch.randn(Co, generator=g(3))

    def ref(x, W, b):
        y = F.conv_transpose2d(x.permute(0, 3, 1, 2), W, b, stride=2)
        return y if nchw else y.permute(0, 2, 3, 1)

    def hip(x, W, b):
        return ops.convT2d_s2(x, W, b, out_nchw=nchw)
    assert ops.CONVT_DIRECT
    monkeypatch.setattr(ops, 'CONVT_DIRECT_BWD', bwd)       # (the direct backward kernels are opt-in: measured slower; parity is pinned all the same)
    y = hip(x.cuda(), W.cuda(), b.cuda())
    r = ref(x.double(), W.double(), b.double())
    assert y.shape == r.shape
    err = (y.cpu().double() - r).abs().max().item() / r.abs().max().item()
    assert err < 2e-6, err
    _check(hip, ref, [x, W, b], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('N,Hi,Wi', [(8, 64, 64), (3, 64, 64), (2, 128, 128), (5, 34, 30), (4, 20, 66), (1, 6, 4), (33, 64, 64)])
def test_first_layer_straight_from_the_u8_frames(N, Hi, Wi, monkeypatch):
    """genrl_conv1_u8_fwd / genrl_conv1_u8_wgrad (the first encoder layer without a patch matrix, ops_conv_planes._conv1_direct) against torch's
    conv2d on x / 255 - 0.5 in float64 + LayerNorm + SiLU -- output, weight / bias / LayerNorm gradients -- and against the im2col + GEMM
    path it replaces (GENRL_CONV1_DIRECT=0) at fp32 rounding; row widths that end inside a 16-pixel block (Wo = 31, 63, 14), one that fills
    its blocks exactly (Wo = 32), a single-pixel row"""
    from genrl_amd import ops_conv_planes as cp
    x = torch.randint(0, 256, (N, 3, Hi, Wi), generator=g(1), dtype=torch.uint8)
    W = torch.randn(48, 3, 4, 4, generator=g(2)) / 48 ** .5; b = 0.1 * torch.randn(48, generator=g(3))
    ga, be = 1 + 0.1 * torch.randn(48, generator=g(4)), 0.1 * torch.randn(48, generator=g(5))
    Ho, Wo = (Hi - 4) // 2 + 1, (Wi - 4) // 2 + 1
    wgt = torch.randn(N, Ho, Wo, 48, generator=g(6))

    def run(direct):
        monkeypatch.setattr(cp, 'CONV1_DIRECT', direct)
        ps = [t.clone().cuda().requires_grad_(True) for t in (W, b, ga, be)]
        y = cp.conv2d_s2(x.cuda(), ps[0], ps[1], (ps[2], ps[3], 1e-3), fp32_out=True)
        (y * wgt.cuda()).sum().backward()
        return [y.detach()] + [p.grad.detach() for p in ps]
    new, old = run(True), run(False)
    xd = (x.double() / 255.0 - 0.5)
    ps = [t.clone().double().requires_grad_(True) for t in (W, b, ga, be)]
    ref = _ln_silu(F.conv2d(xd, ps[0], ps[1], stride=2).permute(0, 2, 3, 1), ps[2], ps[3])
    (ref * wgt.double()).sum().backward()
    refs = [ref.detach()] + [p.grad for p in ps]
    for name, a, o, rf in zip(('y', 'dW', 'db', 'dgamma', 'dbeta'), new, old, refs):
        scale = max(1e-6, rf.abs().max().item())
        assert a.shape == rf.shape and torch.isfinite(a).all(), name
        e_new = (a.cpu().double() - rf).abs().max().item() / scale
        e_old = (o.cpu().double() - rf).abs().max().item() / scale
        assert e_new <= 2e-5, (name, e_new, e_old)
        assert e_new <= 4 * e_old + 2e-6, (name, e_new, e_old)        # no worse than the path it replaces (fp32 summation orders differ)
