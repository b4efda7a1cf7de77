"""The stride-2 convolutions of the image encoder / decoder (agent/dreamer_utils.py:558-715) with their products on h2 planes.

Same autograd node structure as ops._Conv2dS2 / ops._ConvT2dS2 (NHWC activations, channel-LayerNorm + SiLU inside the node, the
scatter side as GEMM -> col2im), but every product that can runs on the fp16 matrix cores through the plane kernels:

  forward conv / input gradient of the transposed conv   patch-gathering plane GEMM          genrl_gemm_h2_conv
  forward transposed conv / input gradient of the conv    plain plane GEMM (rows = pixels)     genrl_gemm_h2      -> col2im
  weight gradients (sum over pixels)                      transposing plane GEMM + gather      genrl_gemm_h2_tn_conv

A patch row gathers from k x k pixel rows and a weight gradient sums over pixels, so the gathered / summed operands carry ONE scale
for the whole tensor ("uniform" planes): activations get it from the channel-LayerNorm kernel itself (the scale its parameters
guarantee, genrl_ln_act_fwd_h2u), gradients from the LayerNorm backward's partial maxima (genrl_ln_act_bwd_h2u), anything else
from genrl_split_h2u.  Products whose shapes the plane kernels do not take (3-channel ends, pixel counts that are not a multiple of
64, fewer than MIN_ROWS rows) stay on the fp32-operand kernels of ops.py -- decided per product."""
import os
import torch
from torch.autograd import Function

from ._lib import lib, check
from . import planes
from . import ops
from .ops import _p, _f32, _grad_buf, _ws, sgemm, sgemm_conv, colsum

_stream = ops._stream
ENABLED = os.environ.get('GENRL_PLANES_CONV', '1') != '0'


def min_rows():
    """pixel rows from which a convolution product takes the plane kernels (GENRL_PLANES_CONV_MIN_ROWS overrides)"""
    return int(os.environ.get('GENRL_PLANES_CONV_MIN_ROWS', '4096'))


def active(x):
    return ENABLED and planes.ENABLED and x.is_cuda


_rowoff_cache = {}


def _rowoff(Nimg, H, W, k, ld, dev):
    """byte offset of the first pixel row of patch m = (image, oy, ox) in planes [pixel][ld]: ((n H + 2 oy) W + 2 ox) ld 2
    (uint32 as int32 bits; M + 256 entries, the tail repeats the last value: the kernel's run-ahead reads there)"""
    key = (Nimg, H, W, k, ld, str(dev))
    t = _rowoff_cache.get(key)
    if t is None:
        Ho, Wo = (H - k) // 2 + 1, (W - k) // 2 + 1
        n = torch.arange(Nimg, device=dev, dtype=torch.int64)[:, None, None]
        oy = torch.arange(Ho, device=dev, dtype=torch.int64)[None, :, None]
        ox = torch.arange(Wo, device=dev, dtype=torch.int64)[None, None, :]
        off = (((n * H + 2 * oy) * W + 2 * ox) * ld * 2).reshape(-1)
        assert int(off[-1]) + 2 * ld * (k * W + k) < 2 ** 32
        off = torch.cat([off, off[-1:].expand(256)])
        t = (off & 0xFFFFFFFF).to(torch.int64)
        t = torch.where(t >= 2 ** 31, t - 2 ** 32, t).to(torch.int32).contiguous()
        _rowoff_cache[key] = t
    return t


def _uniform_split(x2d):
    """uniform-scale planes of an fp32 matrix (exact tensor maximum; two launches)"""
    R, C = x2d.shape
    P = planes.Planes(R, C, x2d.device, zero=False)
    ws = torch.empty(1024, device=x2d.device)
    check(lib().genrl_split_h2u(_p(x2d), C, R, C, P.ptr(), P.ld, P.plane, P.inv_ptr(), _p(ws), _stream()), 'split_h2u')
    P.uniform = True
    return P


LAZY_FP32 = os.environ.get('GENRL_CONV_LAZY_FP32', '1') != '0'
CONV1_DIRECT = os.environ.get('GENRL_CONV1_DIRECT', '1') != '0'
KEEP_COLS = True      # (where the first layer still runs im2col + GEMM) the u8 frames' patch matrix is kept for its weight gradient: one im2col launch less


def _ln_fwd(pre2d, gamma, beta, eps, want_planes, want_fp32=True):
    """channel-LayerNorm + SiLU of the rows; -> (y, mean, rstd, uniform planes of y or None, lazy).  want_fp32 False (an inner layer whose
    consumer gathers from the planes): y is allocated but NOT written when the kernel makes the planes itself -- lazy is then a cell
    [True] and a consumer that does read fp32 values after all fills y from the planes first (_need_fp32)"""
    M, N = pre2d.shape
    y = torch.empty_like(pre2d)
    mean = torch.empty(M, device=pre2d.device); rstd = torch.empty(M, device=pre2d.device)
    P = lazy = None
    if want_planes and N <= 256 and N % 4 == 0 and M >= 64:
        P = planes.Planes(M, N, pre2d.device, zero=False)
        if not want_fp32 and LAZY_FP32:
            lazy = [True]
        check(lib().genrl_ln_act_fwd_h2u(_p(pre2d), N, _p(gamma), _p(beta), None if lazy else _p(y), N, _p(mean), _p(rstd), M, N, eps, 1,
                                         P.ptr(), P.ld, P.plane, P.inv_ptr(), _stream()), 'ln_act_fwd_h2u')
        P.uniform = True
    else:
        check(lib().genrl_ln_act_fwd(_p(pre2d), N, _p(gamma), _p(beta), _p(y), N, _p(mean), _p(rstd), M, N, eps, 1, _stream()),
              'ln_act_fwd')
        if want_planes and N % 4 == 0:
            P = _uniform_split(y)
    return y, mean, rstd, P, lazy


def _need_fp32(x, cell, xp):
    """x's fp32 values are about to be read: fill them from the planes if the producer skipped them (see _ln_fwd).  One HBM pass of a
    dedicated kernel (genrl_planes_to_f32: no float64, no temporaries -- inside a hipGraph capture temporaries would stay in the graph's pool)"""
    if cell is not None and cell[0]:
        assert x.is_contiguous() and x.numel() == xp.rows * xp.cols
        # (x.data_ptr(): the values autograd saw were never defined -- written behind its back, no version bump)
        check(lib().genrl_planes_to_f32(xp.ptr(), xp.ld, xp.plane, xp.inv_ptr(), x.data_ptr(), xp.cols, xp.rows, xp.cols, _stream()), 'planes_to_f32')
        cell[0] = False


def _ln_bwd(dy2d, pre2d, gamma, beta, mean, rstd, bias, want_planes):
    """-> dpre, dgamma, dbeta, dbias (None where accumulated straight into the flat gradient buffers), uniform planes of dpre"""
    M, N = pre2d.shape
    dev = pre2d.device
    dpre = torch.empty_like(pre2d)
    tg, tb, tc = _grad_buf(gamma), _grad_buf(beta), _grad_buf(bias)
    direct = tg is not None and tb is not None and tc is not None
    if direct:
        g0, g1, g2 = tg, tb, tc
    else:
        gb = torch.empty(3, N, device=dev)
        g0, g1, g2 = gb[0], gb[1], gb[2]
    ws = _ws(lib().genrl_ln_ws_floats(M, N), dev)
    acc_p = int(direct) | (ops.defer_reduce(M, N, ws, g0, g1, g2) if direct else 0)
    P = None
    if want_planes and N <= 256 and N % 4 == 0 and M >= 64:
        P = planes.Planes(M, N, dev, zero=False)
        amax = torch.empty(2048, device=dev)
        check(lib().genrl_ln_act_bwd_h2u(_p(dy2d), N, _p(pre2d), N, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dpre), N, _p(g0), _p(g1),
                                         _p(g2), _p(ws), M, N, 1, acc_p, P.ptr(), P.ld, P.plane, P.inv_ptr(), _p(amax), _stream()),
              'ln_act_bwd_h2u')
        P.uniform = True
    else:
        check(lib().genrl_ln_act_bwd(_p(dy2d), N, _p(pre2d), N, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dpre), N, _p(g0), _p(g1),
                                     _p(g2), _p(ws), M, N, 1, acc_p, _stream()), 'ln_act_bwd')
        if want_planes and N % 4 == 0:
            P = _uniform_split(dpre)
    return (dpre, None, None, None, P) if direct else (dpre, g0, g1, g2, P)


def _gemm_conv(img_p, Nimg, H, W, C, k, Bp, out, ldc, bias, N):
    """out[m, n] = sum_kk patch(m, kk) B[n, kk] (+ bias): the DMA gathers the patches from the uniform planes img_p"""
    if planes.gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    check(lib().genrl_gemm_h2_conv(img_p.ptr(), img_p.ld, img_p.plane, img_p.inv_ptr(), Nimg, H, W, C, k, Bp.ptr(), Bp.ld, Bp.plane,
                                   Bp.inv_ptr(), _p(out), ldc, _p(bias), N, 0, _stream()), 'gemm_h2_conv')
    if planes.gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        Ho, Wo = (H - k) // 2 + 1, (W - k) // 2 + 1
        planes.gemm_profile.append((Nimg * Ho * Wo, N, k * k * C, e0, e1, 'kk/conv1/h2/pipe4'))


def _gemm_tn_conv(Ap, img_p, Nimg, H, W, C, k, out, ldc, NI, M):
    """out[i, j] = sum_m A(m, i) patch(m, j): the convolution weight gradients (A: planes with M rows; patches from img_p)"""
    if planes.gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    NJ = k * k * C
    nb = lib().genrl_gemm_h2_tn_ws_bytes(NI, NJ, M)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=out.device)
    wp = (ws.data_ptr() + 255) // 256 * 256
    ro = _rowoff(Nimg, H, W, k, img_p.ld, out.device)
    assert ro.numel() == M + 256, (ro.numel(), M)
    check(lib().genrl_gemm_h2_tn_conv(Ap.ptr(), Ap.ld, Ap.plane, Ap.inv_ptr(), img_p.ptr(), img_p.ld, img_p.plane, img_p.inv_ptr(),
                                      _p(ro), W, C, k, _p(out), ldc, NI, M, 0, wp, nb, _stream()), 'gemm_h2_tn_conv')
    if planes.gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        planes.gemm_profile.append((NI, NJ, M, e0, e1, 'rr/conv2/h2tn/pipe4'))


# ---- gather ("sub-pixel") form of the scatter side: ConvTranspose2d forward / Conv2d input gradient without cols + col2im ---------
SUBPIXEL = os.environ.get('GENRL_SUBPIXEL', '1') != '0'
# odd kernels (k = 5 treated as 6 with a zero tap: 1.4 x the taps).  Round 4 measured it neutral on c2's 5^2 -> 13^2 layer and left it off; with the
# 128 x 192 tile and at c4's sizes (the 29^2 -> 61^2 layer: a 1.66 GB cols matrix otherwise) it pays: c4 40.3 -> 39.5 ms, c3 37.0 -> 36.75, c2 22.0 -> 22.0
# (round 5, scripts/r05_odd_ab.sh): on by default, GENRL_SUBPIXEL_ODD=0 restores GEMM -> col2im for odd kernels
SUBPIXEL_ODD = os.environ.get('GENRL_SUBPIXEL_ODD', '1') != '0'


def _hl_on():
    return os.environ.get('GENRL_PLANES_HL', '1')[:1] != '0'


def _subpixel_dims_ok(Nimg, Hi, Wi, Cs, Cp, k):
    """the shape side of subpixel_ok (what a PRODUCER can know about its consumer before the planes exist)"""
    T = (k + 1) // 2
    if not (SUBPIXEL and _hl_on() and (k % 2 == 0 or SUBPIXEL_ODD) and Cs % 8 == 0 and Cs >= 48 and Cp % 4 == 0
            and T * T * Cs >= 64 and Nimg * (Hi + T - 1) * (Wi + T - 1) >= min_rows() // 4):
        return False
    # the kernel's own size limits (genrl_gemm_h2_subpixel: 32-bit byte offsets into the padded planes, 31-bit row count): a layer beyond
    # them -- larger frames at a larger per-GPU batch -- keeps the GEMM -> col2im form instead of failing
    Hp, Wp = Hi + 2 * (T - 1), Wi + 2 * (T - 1)
    ld = (Cs + 63) // 64 * 64
    plane = Nimg * Hp * Wp * ld
    return (Nimg * Hp * Wp * ld + plane) * 2 < 0xffffffff and Nimg * (Hp - T + 1) * (Wp - T + 1) <= 0x7fffffff


def subpixel_ok(xp, Nimg, Hi, Wi, Cs, Cp, k):
    """may genrl_gemm_h2_subpixel take this layer?  xp: planes of the [Nimg Hi Wi][Cs] input (must be uniform-scale); Cs summed
    channels, Cp produced channels, k the stride-2 kernel"""
    return xp is not None and xp.uniform and xp.cols == Cs and _subpixel_dims_ok(Nimg, Hi, Wi, Cs, Cp, k)


def consumer_reads_planes_only(consumer, Nimg, H, W, C):
    """Will the NEXT layer read this layer's (Nimg, H, W, C) output from its uniform planes alone -- forward product AND weight gradient --
    so that the channel-LayerNorm need not write the fp32 activation at all?  consumer: ('conv', k) = a stride-2 convolution of this module,
    ('convT', Co, k) = a stride-2 transposed convolution.  The very predicates the consumers apply (a wrong 'yes' is still correct: the
    consumer fills the fp32 values from the planes, _need_fp32 -- one extra pass; a wrong 'no' writes an activation nobody reads)."""
    if not consumer or not LAZY_FP32:
        return False
    if consumer[0] == 'conv':
        k = consumer[1]
        Ho, Wo = (H - k) // 2 + 1, (W - k) // 2 + 1
        M = Nimg * Ho * Wo
        return Ho > 0 and Wo > 0 and _gather_ok(M, C) and M % 64 == 0 and (C * k * k) % 8 == 0
    if consumer[0] == 'convT':
        _, Co, k = consumer
        M = Nimg * H * W
        on_planes = M * k * k >= min_rows()
        fwd = on_planes and ((H > 1 and _subpixel_dims_ok(Nimg, H, W, C, Co, k)) or C >= KR_MIN_K)
        tn = _gather_ok(M, Co, k) and C % 4 == 0 and M % 64 == 0
        return fwd and tn
    return False


def _subpixel(xp, Nimg, Hi, Wi, Cs, Cp, k, Wsrc, s_ci, s_co, s_tap, bias, out, wkey=None):
    """out[n][2 py + a][2 px + b][cp] = bias[cp] + sum over the T x T patch of the zero-padded input and the Cs summed channels
    (genrl_gemm_h2_subpixel); out: fp32 NHWC (Nimg, Ho, Wo, Cp), fully written when Ho <= 2 (Hi + T - 1) (the caller zero-fills otherwise)"""
    T = (k + 1) // 2
    pad = T - 1
    dev = out.device
    Hp, Wp = Hi + 2 * pad, Wi + 2 * pad
    if planes.gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    xq = planes.Planes(Nimg * Hp * Wp, Cs, dev, zero=False)
    assert xq.ld == xp.ld
    check(lib().genrl_pad_planes(xp.ptr(), xp.plane, xp.inv_ptr(), xq.ptr(), xq.plane, xq.inv_ptr(), Nimg, Hi, Wi, xp.ld, pad, _stream()),
          'pad_planes')
    K = T * T * Cs

    def build():
        # (from the parameter alone when the caller names it: Wsrc is
        # then this iteration's permuted copy, ops.permuted(wkey) the current one, same layout)
        src = ops.permuted(wkey) if isinstance(wkey, torch.nn.Parameter) else Wsrc
        wsub = torch.empty(4 * Cp, K, device=dev)
        b4_ = torch.empty(4 * Cp, device=dev) if bias is not None else None
        check(lib().genrl_subpixel_weight(_p(src), s_ci, s_co, s_tap, Cs, Cp, k, T, _p(wsub), _p(bias), _p(b4_), _stream()), 'subpixel_weight')
        return planes.split(wsub), b4_
    # (once per optimiser step when the caller names the parameter the weight comes from; weight and bias are stepped together)
    # (a bias edited or swapped without its weight rebuilds too: its identity + version are part of the entry's freshness check, not of its key)
    bver = (id(bias), bias._version) if bias is not None else None
    wp, b4 = planes.derived(wkey, 'subpixel', build, extra=bver) if wkey is not None else build()
    _, Ho, Wo, _ = out.shape
    check(lib().genrl_gemm_h2_subpixel(xq.ptr(), xq.ld, xq.plane, xq.inv_ptr(), Nimg, Hp, Wp, Cs, T, wp.ptr(), wp.ld, wp.plane, wp.inv_ptr(),
                                       _p(out), Ho, Wo, Cp, _p(b4), _stream()), 'gemm_h2_subpixel')
    if planes.gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        planes.gemm_profile.append((Nimg * (Hp - T + 1) * (Wp - T + 1), 4 * Cp, K, e0, e1, 'kk/subpixel/h2/pipe4'))


def _wplanes(wsrc, Wp, transpose):
    """planes of the permuted weight matrix Wp (or of its transpose): once per optimiser step when the parameter it comes from is known"""
    if wsrc is None:
        return planes.split(Wp.detach(), transpose=transpose)
    shape = tuple(Wp.shape)
    src = (lambda: ops.permuted(wsrc).reshape(shape)) if isinstance(wsrc, torch.nn.Parameter) else (lambda: Wp.detach())
    return planes.derived(wsrc, ('perm_planes', transpose), lambda: planes.split(src(), transpose=transpose))


KR_MIN_K = int(os.environ.get('GENRL_PLANES_KR_MIN_K', '512'))      # GEMM -> col2im products: with few input channels (K = C) a 128 x 128 tile is two K stages of prologue and a
#                     64 KiB store -- the write-bound fp32-operand kernel is faster there (173056 x 1728 x 96: 0.55 vs 1.0 ms)


def _conv1_direct(x, C, k, Co, Wi):
    """may genrl_conv1_u8_fwd / _wgrad take the first encoder layer?  (precision 16 rounds operands in the GEMM kernels proper: it keeps im2col + GEMM)"""
    return CONV1_DIRECT and C == 3 and k == 4 and Co == 48 and Wi % 2 == 0 and x.data_ptr() % 2 == 0 and not ops._p16()


def _gather_ok(M, C, k=1):
    return M * k * k >= min_rows() and C % 8 == 0 and C >= 48


class _Conv2dS2P(Function):
    """ops._Conv2dS2 with plane products.  x: f32 NHWC (N,H,W,C) [xp: its uniform planes or None] or u8 NCHW frames;
    Wp (Co, k*k*Ci) = weight permuted to (co, kh, kw, ci); channel-LayerNorm + SiLU fused; returns NHWC with ._planes set."""
    @staticmethod
    def forward(ctx, x, Wp, b, k, gamma, beta, eps, xp, holder, wsrc=(None,), opts=(None, True, True)):
        ctx.wsrc = wsrc[0]
        ctx.xlazy, fp32_out, planes_out = opts      # (cell of a planes-only input; does anything read this layer's fp32 output / planes?)
        u8 = x.dtype == torch.uint8
        x = x.contiguous()
        if u8:
            Nimg, C, Hi, Wi = x.shape
        else:
            Nimg, Hi, Wi, C = x.shape
        Co = Wp.shape[0]
        Ho, Wo = (Hi - k) // 2 + 1, (Wi - k) // 2 + 1
        M, K = Nimg * Ho * Wo, C * k * k
        y = torch.empty(M, Co, device=x.device)
        on_planes = (not u8) and xp is not None and _gather_ok(M, C)
        if on_planes:
            _gemm_conv(xp, Nimg, Hi, Wi, C, k, _wplanes(ctx.wsrc, Wp, False), y, Co, b, Co)
        elif u8 and _conv1_direct(x, C, k, Co, Wi):
            # the first layer straight from the frames (genrl_conv1_u8_fwd): no patch matrix, forward or backward
            if planes.gemm_profile is not None:
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
            check(lib().genrl_conv1_u8_fwd(_p(x), _p(Wp), _p(b), _p(y), Nimg, Hi, Wi, Co, k, _stream()), 'conv1_u8_fwd')
            if planes.gemm_profile is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                planes.gemm_profile.append((M, Co, K, e0, e1, 'kk/conv1_direct'))
            ctx.direct1 = True
        else:
            _need_fp32(x, ctx.xlazy, xp)
            if ops._implicit_conv(x, C):
                sgemm_conv(x, K, 1, Wp, K, 1, y, Co, b, M, Co, K, 1, (Hi, Wi, C, k))
            else:
                cols = ops._im2col(x, Nimg, Hi, Wi, C, k, 2 if u8 else 0)
                sgemm(cols, K, 1, Wp, K, 1, y, Co, b, M, Co, K)
                if u8 and KEEP_COLS and ctx.needs_input_grad[1]:
                    ctx.cols = cols          # (the frames' patch matrix serves the weight gradient again: 0.15 GB at c2, 0.8 GB at c4, of 288)
        if isinstance(fp32_out, tuple):       # the consumer's description: decided by ITS predicates, not by the layer's position
            fp32_out = not consumer_reads_planes_only(fp32_out, Nimg, Ho, Wo, Co)
        out, mean, rstd, outp, lazy = _ln_fwd(y, gamma, beta, eps, want_planes=planes_out and M >= min_rows(), want_fp32=fp32_out)
        holder.append(outp); holder.append(lazy)
        ctx.dims = (Nimg, Hi, Wi, C, k, u8)
        ctx.bias = b
        ctx.xp = xp if on_planes else None
        ctx.save_for_backward(x, Wp, y, mean, rstd, gamma, beta)
        return out.reshape(Nimg, Ho, Wo, Co)

    @staticmethod
    def backward(ctx, dy):
        x, Wp, pre, mean, rstd, gamma, beta = ctx.saved_tensors
        Nimg, Hi, Wi, C, k, u8 = ctx.dims
        Co = Wp.shape[0]
        K = C * k * k
        Ho, Wo = (Hi - k) // 2 + 1, (Wi - k) // 2 + 1
        M = Nimg * Ho * Wo
        xp = ctx.xp
        need_dx = (not u8) and ctx.needs_input_grad[0]
        tn = xp is not None and ctx.needs_input_grad[1] and M % 64 == 0 and K % 8 == 0
        kr = need_dx and M >= min_rows() and Co >= KR_MIN_K
        # (the LayerNorm backward's planes are uniform-scale whenever it makes them itself: Co <= 256)
        sp_maybe = need_dx and SUBPIXEL and _hl_on() and (k % 2 == 0 or SUBPIXEL_ODD) and Co % 8 == 0 and Co >= 48 and C % 4 == 0
        dy2, dg, dbe, db, dyp = _ln_bwd(dy.reshape(M, Co).contiguous(), pre, gamma, beta, mean, rstd, ctx.bias, want_planes=tn or kr or sp_maybe)
        dx = dW = None
        if ctx.needs_input_grad[1]:
            dW = torch.empty(Co, K, device=dy.device)
            if not (tn and dyp is not None):
                _need_fp32(x, ctx.xlazy, xp)
            if getattr(ctx, 'direct1', False):
                ws = torch.empty(lib().genrl_conv1_u8_wgrad_ws_floats(Co), device=dy.device)
                if planes.gemm_profile is not None:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record()
                check(lib().genrl_conv1_u8_wgrad(_p(x), _p(dy2), _p(dW), _p(ws), Nimg, Hi, Wi, Co, k, _stream()), 'conv1_u8_wgrad')
                if planes.gemm_profile is not None:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record()
                    planes.gemm_profile.append((Co, K, M, e0, e1, 'rr/conv1_direct_bwd'))
            elif tn and dyp is not None:
                _gemm_tn_conv(dyp, xp, Nimg, Hi, Wi, C, k, dW, K, Co, M)
            elif ops._implicit_conv(x, C) and Co % 4 == 0:
                sgemm_conv(dy2, 1, Co, x, 1, K, dW, K, None, Co, K, M, 2, (Hi, Wi, C, k))
            else:
                cols = getattr(ctx, 'cols', None)
                if cols is None:
                    cols = ops._im2col(x, Nimg, Hi, Wi, C, k, 2 if u8 else 0)
                sgemm(dy2, 1, Co, cols, 1, K, dW, K, None, Co, K, M)
                ctx.cols = None
                del cols
        if need_dx and sp_maybe and subpixel_ok(dyp, Nimg, Ho, Wo, Co, C, k):
            # gather form: dx[n][2 py + a][2 px + b][ci] = sum over the T x T patch of the zero-padded dy and co -- no cols, no col2im
            T = (k + 1) // 2
            full = Hi <= 2 * (Ho + T - 1) and Wi <= 2 * (Wo + T - 1)        # (an odd input size leaves its last row / column without a patch)
            dx = (torch.empty if full else torch.zeros)(Nimg, Hi, Wi, C, device=dy.device)
            _subpixel(dyp, Nimg, Ho, Wo, Co, C, k, Wp, K, 1, C, None, dx, wkey=ctx.wsrc)      # Wp = (co, kh, kw, ci): strides of (co, ci, tap) = (K, 1, C)
        elif need_dx:
            dcols = torch.empty(M, K, device=dy.device)
            if kr and dyp is not None:
                planes.gemm(dyp, _wplanes(ctx.wsrc, Wp, True), dcols, K, None, M, K)      # dcols = dy W
            else:
                sgemm(dy2, Co, 1, Wp, 1, K, dcols, K, None, M, K, Co)
            dx = ops._col2im(dcols, None, Nimg, Ho, Wo, C, k, Hi, Wi)
        return dx, dW, db, None, dg, dbe, None, None, None, None, None


class _ConvT2dS2P(Function):
    """ops._ConvT2dS2 (with the fused channel-LayerNorm + SiLU) with plane products.  x NHWC (N,Hi,Wi,Ci), xp: planes of its rows
    (uniform or per-row scales) or None; Wp (Ci, k*k*Co) = weight permuted to (ci, kh, kw, co)."""
    @staticmethod
    def forward(ctx, x, Wp, b, k, gamma, beta, eps, xp, holder, wsrc=(None,), opts=(None, True, True)):
        ctx.wsrc = wsrc[0]
        ctx.xlazy, fp32_out, planes_out = opts
        if ctx.xlazy is not None and not (x.dtype == torch.float32 and x.is_contiguous()):
            _need_fp32(x, ctx.xlazy, xp)
        x = _f32(x).contiguous()
        Nimg, Hi, Wi, Ci = x.shape
        Nw = Wp.shape[1]
        Co = Nw // (k * k)
        M = Nimg * Hi * Wi
        if xp is None and M >= min_rows() // 4 and Ci % 4 == 0:
            xp = planes.split(x.reshape(M, Ci))                      # (the first layer's input comes from a Linear)
        on_planes = xp is not None and M * (k * k) >= min_rows()
        if on_planes and Hi > 1 and subpixel_ok(xp, Nimg, Hi, Wi, Ci, Co, k):
            # gather form: every output pixel sums its T x T patch of the zero-padded input -- no cols matrix, no col2im
            y = torch.empty(Nimg, 2 * (Hi - 1) + k, 2 * (Wi - 1) + k, Co, device=x.device)
            _subpixel(xp, Nimg, Hi, Wi, Ci, Co, k, Wp, Nw, 1, Co, b, y, wkey=ctx.wsrc)         # Wp = (ci, kh, kw, co): strides of (ci, co, tap) = (Nw, 1, Co)
        else:
            cols = torch.empty(M, Nw, device=x.device)
            if on_planes and Ci >= KR_MIN_K:
                planes.gemm(xp, _wplanes(ctx.wsrc, Wp, True), cols, Nw, None, M, Nw)      # cols = x W
            else:
                _need_fp32(x, ctx.xlazy, xp)
                sgemm(x, Ci, 1, Wp, 1, Nw, cols, Nw, None, M, Nw, Ci)
            y = ops._col2im(cols, b, Nimg, Hi, Wi, Co, k)
            del cols
        Ho, Wo = y.shape[1], y.shape[2]
        if isinstance(fp32_out, tuple):
            fp32_out = not consumer_reads_planes_only(fp32_out, Nimg, Ho, Wo, Co)
        out, mean, rstd, outp, lazy = _ln_fwd(y.reshape(-1, Co), gamma, beta, eps, want_planes=planes_out and Nimg * Ho * Wo >= min_rows(),
                                              want_fp32=fp32_out)
        holder.append(outp); holder.append(lazy)
        ctx.dims = (Nimg, Hi, Wi, Ci, Co, k)
        ctx.bias = b
        ctx.xp = xp if on_planes else None
        ctx.save_for_backward(x, Wp, y, mean, rstd, gamma, beta)
        return out.reshape(y.shape)

    @staticmethod
    def backward(ctx, dy):
        x, Wp, pre, mean, rstd, gamma, beta = ctx.saved_tensors
        Nimg, Hi, Wi, Ci, Co, k = ctx.dims
        Ho, Wo = 2 * (Hi - 1) + k, 2 * (Wi - 1) + k
        M, Nw = Nimg * Hi * Wi, Co * k * k
        xp = ctx.xp
        gather = _gather_ok(M, Co, k) and Ci % 4 == 0
        tn = gather and xp is not None and M % 64 == 0 and ctx.needs_input_grad[1]
        dgr = gather and ctx.needs_input_grad[0]
        dyv, dg, dbe, db, dyp = _ln_bwd(dy.contiguous().reshape(-1, Co), pre.reshape(-1, Co), gamma, beta, mean, rstd, ctx.bias,
                                       want_planes=tn or dgr)
        dyv = dyv.reshape(Nimg, Ho, Wo, Co)
        implicit = ops._implicit_conv(dyv, Co) and Ci % 4 == 0
        dcols = None
        dx = dW = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, Ci, device=dy.device)
            if dgr and dyp is not None:
                _gemm_conv(dyp, Nimg, Ho, Wo, Co, k, _wplanes(ctx.wsrc, Wp, False), dx, Ci, None, Ci)      # dx = patches(dy) W^T
            elif implicit:
                sgemm_conv(dyv, Nw, 1, Wp, Nw, 1, dx, Ci, None, M, Ci, Nw, 1, (Ho, Wo, Co, k))
            else:
                dcols = ops._im2col(dyv, Nimg, Ho, Wo, Co, k, 0)
                sgemm(dcols, Nw, 1, Wp, Nw, 1, dx, Ci, None, M, Ci, Nw)
            dx = dx.reshape(Nimg, Hi, Wi, Ci)
        if ctx.needs_input_grad[1]:
            dW = torch.empty(Ci, Nw, device=dy.device)
            if not (tn and dyp is not None):
                _need_fp32(x, ctx.xlazy, xp)
            if tn and dyp is not None:
                _gemm_tn_conv(xp, dyp, Nimg, Ho, Wo, Co, k, dW, Nw, Ci, M)                              # dW = x^T patches(dy)
            elif implicit:
                sgemm_conv(x, 1, Ci, dyv, 1, Nw, dW, Nw, None, Ci, Nw, M, 2, (Ho, Wo, Co, k))
            else:
                if dcols is None:
                    dcols = ops._im2col(dyv, Nimg, Ho, Wo, Co, k, 0)
                sgemm(x, 1, Ci, dcols, 1, Nw, dW, Nw, None, Ci, Nw, M)
        return dx, dW, db, None, dg, dbe, None, None, None, None, None


def conv2d_s2(x, W, b, ln, fp32_out=True, planes_out=True):
    """ops.conv2d_s2 with the fused channel-LayerNorm; the output carries ._planes (uniform planes of its pixel rows) for the
    next layer when it was worth making them.  fp32_out False: the caller passes the output to another layer of this module only,
    which reads the planes (the fp32 values are then filled on demand: ._lazy); fp32_out = ('conv', k) / ('convT', Co, k): the NEXT
    layer's description -- the fp32 activation is skipped exactly when that layer's own predicates say it reads planes only
    (consumer_reads_planes_only); planes_out False: nothing reads the planes"""
    Co, Ci, k, _ = W.shape
    Wp = ops._PermuteWeight.apply(W).reshape(Co, k * k * Ci)
    holder = []
    y = _Conv2dS2P.apply(x, Wp, b, k, ln[0], ln[1], float(ln[2]), getattr(x, '_planes', None), holder, (W,),
                         (getattr(x, '_lazy', None), fp32_out, planes_out))
    y._planes, y._lazy = holder if holder else (None, None)
    return y


def convT2d_s2(x, W, b, ln, fp32_out=True, planes_out=True):
    Ci, Co, k, _ = W.shape
    Wp = ops._PermuteWeight.apply(W).reshape(Ci, k * k * Co)
    holder = []
    y = _ConvT2dS2P.apply(x, Wp, b, k, ln[0], ln[1], float(ln[2]), getattr(x, '_planes', None), holder, (W,),
                          (getattr(x, '_lazy', None), fp32_out, planes_out))
    y._planes, y._lazy = holder if holder else (None, None)
    return y
