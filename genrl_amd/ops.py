"""torch.autograd bindings of the HIP kernels (libgenrl_hip.so).  PyTorch owns device memory,
streams and the autograd graph; every flop of the hot path runs in the hand-written kernels.
No CPU fallback: calling an op with a non-CUDA tensor raises."""
import os
import ctypes
import torch
from torch.autograd import Function

from ._lib import lib, check, GenrlHipError

UNIMIX = 0.99
gemm_profile = None      # set to a list to record (M, N, K, start_event, end_event) per sgemm launch


from .streams import raw_current_stream as _stream      # raw hipStream_t of torch's current stream (private fast call, public fallback)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise GenrlHipError('genrl_amd ops need tensors on the MI355X (no CPU fallback)')
    assert t.is_contiguous(), 'non-contiguous tensor handed to a HIP op'
    return t.data_ptr()


# how `precision: 32` runs its big-tile GEMMs: 'f32' = fp32 MFMA; 'bf16x3-big' / 'bf16x3' = fp32 operands split exactly into
# three bf16 terms, six bf16 MFMA products, fp32 accumulation (fp32-equivalent error; GENRL_GEMM_MODE=0|2|3 overrides)
F32_MODE = {'0': 'f32', '2': 'bf16x3-big', '3': 'bf16x3'}.get(os.environ.get('GENRL_GEMM_MODE', ''), 'bf16x3-big')


def set_gemm_precision(mode):
    """'f32' (default) or 'bf16': MFMA operands rounded to bf16, fp32 accumulation, fp32 tensors (the reference's
    `precision: 16` mode).  Process-wide; returns the previous mode."""
    codes = {'f32': 0, 'bf16': 1, 'bf16x3-big': 2, 'bf16x3': 3}
    assert mode in codes, mode
    prev = lib().genrl_set_gemm_precision(codes[mode])
    return {v: k for k, v in codes.items()}[prev]


def _p16():
    """precision 16 selected (every matrix product rounds both operands to bf16): the kernels that feed exact fp32 MFMAs
    straight from memory without a rounding step (the direct 3-channel transposed convolution, the K-split weight-streaming slabs of the
    scans' backward, the persistent scan) stand aside for the ones that implement the mode"""
    return lib().genrl_gemm_precision() == 1


def _prows(t):
    """pointer of a 2-D operand whose rows are evenly spaced (unit column stride); the callee gets the spacing"""
    if not t.is_cuda:
        raise GenrlHipError('genrl_amd ops need tensors on the MI355X (no CPU fallback)')
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), 'rows must be unit-stride'
    return t.data_ptr()


def _f32(t):
    assert t.dtype == torch.float32, t.dtype
    return t


direct_grads = False      # True only while Optimizer.__call__ runs loss.backward() (agent/dreamer_utils.Optimizer)

# ---- deferred LayerNorm parameter-gradient reductions.  A LayerNorm backward leaves per-workgroup partial rows of (dgamma, dbeta
# [, column sums of dx]) and sums them with a second launch; under the Optimizer -- where the sums are ADDED into the flat gradient
# buffers and nobody reads them before the backward pass is over -- that second launch is skipped (accumulate_params & 4) and all
# of a pass's partial sets are summed by one launch per 24 at its end (genrl_reduce_params_batch: same arithmetic, same order per
# set; ~40 launches fewer per iteration).
_deferred = None          # list of (ws tensor, parts, N, np, g0, g1, g2) while Optimizer.__call__ runs loss.backward()
DEFER_REDUCTIONS = os.environ.get('GENRL_DEFER_REDUCTIONS', '1') != '0'


def defer_begin():
    global _deferred
    if _deferred:                  # (a pass that never reached its flush nor its abort: nothing of it may be summed later)
        _deferred = None
    _deferred = [] if DEFER_REDUCTIONS else None


def defer_abort():
    """drop the registered partial sets without summing them (the backward pass that registered them raised: a launch that failed
    after its set was registered never wrote its workspace)"""
    global _deferred
    _deferred = None


def defer_reduce(M, N, ws, g0, g1, g2=None):
    """-> 4 (the flag for accumulate_params) when the reduction of this call's partial rows is taken over by defer_flush, else 0"""
    if _deferred is None:
        return 0
    parts = lib().genrl_ln_bwd_parts(M, N)
    if parts <= 0:
        return 0
    _deferred.append((ws, parts, N, 3 if g2 is not None else 2, g0, g1, g2))
    return 4


class _ReduceDesc(ctypes.Structure):          # genrl_reduce_desc (include/genrl_hip.h)
    _fields_ = [('part', ctypes.c_void_p), ('out0', ctypes.c_void_p), ('out1', ctypes.c_void_p), ('out2', ctypes.c_void_p),
                ('nchunk', ctypes.c_int), ('N', ctypes.c_int), ('np', ctypes.c_int), ('accumulate', ctypes.c_int)]


def defer_flush():
    """sum every partial set registered since defer_begin into its gradient buffers (on the current stream) and stop deferring"""
    global _deferred
    items, _deferred = _deferred, None
    if not items:
        return
    # a parameter set used more than once in the pass (a shared layer) has several partial sets adding into the SAME buffers:
    # those go into successive launches (within one launch every output has exactly one writer), in registration order
    rounds = []
    for it in items:
        key = it[4].data_ptr()
        for r in rounds:
            if key not in r[0]:
                break
        else:
            r = (set(), [])
            rounds.append(r)
        r[0].add(key); r[1].append(it)
    cur = torch.cuda.current_stream()
    for _, its in rounds:
        for it in its:                 # a workspace filled on a side stream (the connector's) and summed here: keep it alive for this stream
            it[0].record_stream(cur)
        arr = (_ReduceDesc * len(its))()
        for d, (ws, parts, N, np_, g0, g1, g2) in zip(arr, its):
            d.part, d.out0, d.out1, d.out2 = ws.data_ptr(), g0.data_ptr(), g1.data_ptr(), (g2.data_ptr() if g2 is not None else None)
            d.nchunk, d.N, d.np, d.accumulate = parts, N, np_, 1
        check(lib().genrl_reduce_params_batch(arr, len(its), _stream()), 'reduce_params_batch')


def _grad_buf(p):
    """The persistent gradient buffer of a leaf parameter (a view into its optimiser group's flat
    gradient, agent/dreamer_utils.FlatGroup) — weight-gradient GEMMs accumulate straight into it
    (`accumulate=True` epilogue) instead of returning a fresh tensor for autograd to add: no extra
    allocation, no elementwise add per use of a shared weight (16 uses per imagination rollout)."""
    # Only under the Optimizer: torch.autograd.grad, a second backward() or gradient hooks must see the real dW (the
    # backward passes then return it like any autograd node).
    if not direct_grads or p is None or not p.is_leaf or not p.requires_grad:      # (a frozen parameter's buffer must stay untouched)
        return None
    g = p.grad
    if g is not None and g.is_cuda and g.is_contiguous() and g.shape == p.shape:
        return g
    return None


class _WgradStream:
    """Weight-gradient GEMMs are off the critical path of backpropagation (nothing downstream in
    the backward pass reads dW) and accumulate into persistent buffers: when enabled they are
    enqueued on a side stream so that they run beside the dgrad chain (two GEMM kernels per CU =
    two independent workgroups, and their fixed launch/prologue/epilogue costs overlap).  The
    optimiser joins the stream before it reads the gradients.  Operands are kept referenced until
    the join so the caching allocator cannot recycle them under the side stream.
    Measured on MI355X (round 1): no gain over in-order launches (49.65 vs 49.70 ms/step) and the
    fork-per-GEMM pattern does not survive hipGraph capture, so it stays disabled."""
    enabled = False
    keep = []

    @classmethod
    def run(cls, fn, *tensors):
        if not cls.enabled:
            return fn()
        from . import streams
        cls.keep.extend(tensors)
        with streams.fork('wgrad'):
            fn()

    @classmethod
    def join(cls):
        if cls.keep:
            from . import streams
            streams.join('wgrad')
            cls.keep.clear()


wgrad_stream = _WgradStream


def _ws(n, dev):
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=dev)


# ------------------------------------------------------------------ raw (non-autograd) calls

def sgemm(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, accumulate=False, a_off=0, b_off=0, c_off=0):
    """C[m,n] (+)= sum_k A[m*a_rs+k*a_ks] B[n*b_rs+k*b_ks] (+bias[n]); offsets in elements."""
    if gemm_profile is not None:        # bench.py: HIP events around every launch of the GEMM kernel
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    nws = lib().genrl_sgemm_ws_floats(M, N, K)
    ws = torch.empty(nws, dtype=torch.float32, device=C.device) if nws > 0 else None
    check(lib().genrl_sgemm(A.data_ptr() + 4 * a_off, a_rs, a_ks, B.data_ptr() + 4 * b_off, b_rs, b_ks,
                            C.data_ptr() + 4 * c_off, ldc, _p(bias), M, N, K, int(accumulate), _p(ws), nws,
                            _stream()), 'sgemm')
    if gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        gemm_profile.append((M, N, K, e0, e1, ('k' if a_ks == 1 else 'r') + ('k' if b_ks == 1 else 'r') +
                             ('/skinny' if (M <= 32 and a_ks == 1) else f'/pipe{lib().genrl_sgemm_last_pipe()}')))


def sgemm_conv(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, which, img, accumulate=False):
    """sgemm with operand `which` (1 = A, 2 = B) read as the implicit stride-2 patch matrix of the NHWC
    image img = (H, W, C, k) (include/genrl_hip.h: genrl_sgemm_conv) — no materialised im2col."""
    if gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    nws = lib().genrl_sgemm_ws_floats(M, N, K)
    ws = torch.empty(nws, dtype=torch.float32, device=C.device) if nws > 0 else None
    H, W, Cc, k = img
    check(lib().genrl_sgemm_conv(A.data_ptr(), a_rs, a_ks, B.data_ptr(), b_rs, b_ks, C.data_ptr(), ldc, _p(bias),
                                 M, N, K, int(accumulate), _p(ws), nws, which, H, W, Cc, k, _stream()), 'sgemm_conv')
    if gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        gemm_profile.append((M, N, K, e0, e1, ('k' if a_ks == 1 else 'r') + ('k' if b_ks == 1 else 'r') +
                             f'/conv{which}/pipe{lib().genrl_sgemm_last_pipe()}'))


def colsum(x2d, out=None, accumulate=False, ld=None):
    M, N = x2d.shape
    out = out if out is not None else torch.empty(N, device=x2d.device)
    ws = _ws(lib().genrl_colsum_ws_floats(M, N), x2d.device)
    check(lib().genrl_colsum(_prows(x2d), ld or N, _p(out), _p(ws), M, N, int(accumulate), _stream()), 'colsum')
    return out


def copy2d(src, lds, dst, ldd, rows, cols, rowscale=None, accumulate=False, src_off=0, dst_off=0):
    check(lib().genrl_copy2d(src.data_ptr() + 4 * src_off, lds, dst.data_ptr() + 4 * dst_off, ldd, rows, cols,
                             _p(rowscale), int(accumulate), _stream()), 'copy2d')


def cat_cols(parts, rowscale=None):
    """torch.cat(parts, -1) for 2-D row-major parts via strided copies (optionally row-masked)."""
    R = parts[0].shape[0]
    W = sum(p.shape[1] for p in parts)
    out = torch.empty(R, W, device=parts[0].device)
    off = 0
    for p in parts:
        copy2d(p.contiguous(), p.shape[1], out, W, R, p.shape[1], rowscale, dst_off=off)
        off += p.shape[1]
    return out


def onehot_mode(logits):
    """OneHotDist.mode() forward (argmax of the unimix probabilities), no gradient."""
    lg = logits.contiguous()
    K = lg.shape[-1]
    out = torch.empty_like(lg)
    check(lib().genrl_onehot_fwd(_p(lg), None, _p(out), None, lg.numel() // K, K, UNIMIX, _stream()), 'onehot_fwd')
    return out


def align_index(ct, ca, nf):
    """ct (nf.., N, E) target projections (first nf steps used), ca (T, N, E) agent projections ->
    flat target row index (T, N) int64 (tools/genrl_utils.py:344-361)."""
    T, N, E = ca.shape
    urow = torch.empty(T, N, dtype=torch.int64, device=ca.device)
    check(lib().genrl_align_index(_p(ct.contiguous()), _p(ca.contiguous()), _p(urow), T, N, E, nf, _stream()), 'align')
    return urow


def transpose_last2_raw(x, out=None, accumulate=False):
    B, P, C = x.shape
    out = out if out is not None else torch.empty(B, C, P, device=x.device)
    check(lib().genrl_transpose_last2(_p(x.contiguous()), _p(out), B, P, C, int(accumulate), _stream()), 'transpose')
    return out


# ------------------------------------------------------------------ Linear

_NO_PAD = False      # (calibration only: two-hot head rows left unpadded)


def _rows_ld(t2d):
    """Row-major 2-D operand for the GEMMs without a copy: (tensor, leading dimension).  A column slice of a
    wider buffer (rows 16-byte aligned, spaced by a multiple of 4 floats) is used in place."""
    if t2d.stride(1) == 1 and t2d.stride(0) >= t2d.shape[1] and t2d.stride(0) % 4 == 0 and t2d.data_ptr() % 16 == 0:
        return t2d, t2d.stride(0)
    t2d = t2d.contiguous()
    return t2d, t2d.shape[1]


class _Linear(Function):
    """y = x W^T + b.  For N % 4 != 0 (the 255-bin two-hot heads) y is a column slice of a buffer with rows padded
    to a multiple of 4 floats, and a gradient arriving in such a slice is read in place: the vector-load GEMM
    kernels need 16-byte aligned rows (a 255-wide operand falls back to the scalar-load kernel, ~3x slower)."""
    @staticmethod
    def forward(ctx, x, W, b):
        x2 = _f32(x).reshape(-1, x.shape[-1]).contiguous()
        M, K = x2.shape
        N = W.shape[0]
        Np = (N + 3) // 4 * 4 if (N >= 128 and not _NO_PAD) else N   # (narrow heads take the thin-product paths anyway)
        y = torch.empty(M, Np, device=x.device)
        sgemm(x2, K, 1, W, K, 1, y, Np, b, M, N, K)
        ctx.save_for_backward(x2, W)
        ctx.has_bias = b is not None
        ctx.bias = b
        ctx.xshape = x.shape
        return (y if Np == N else y[:, :N]).view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        M, K = x2.shape
        N = W.shape[0]
        dy2, ldy = _rows_ld(dy.reshape(M, N))
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device)
            sgemm(dy2, ldy, 1, W, 1, K, dx, K, None, M, K, N)          # dx = dy W
            dx = dx.reshape(ctx.xshape)
        if ctx.needs_input_grad[1]:
            tgt = _grad_buf(W)
            if tgt is not None:
                wgrad_stream.run(lambda: sgemm(dy2, 1, ldy, x2, 1, K, tgt, K, None, N, K, M, accumulate=True), dy2, x2)
            else:
                dW = torch.empty(N, K, device=dy.device)
                sgemm(dy2, 1, ldy, x2, 1, K, dW, K, None, N, K, M)     # dW = dy^T x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            tgt = _grad_buf(ctx.bias)
            if tgt is not None:
                colsum(dy2, out=tgt, accumulate=True, ld=ldy)
            else:
                db = colsum(dy2, ld=ldy)
        return dx, dW, db


def linear(x, W, b=None):
    return _Linear.apply(x, W, b)


# ------------------------------------------------------------------ LayerNorm (+SiLU)

class _LNAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, act):
        x2 = _f32(x).reshape(-1, x.shape[-1]).contiguous()
        M, N = x2.shape
        y = torch.empty_like(x2)
        mean = torch.empty(M, device=x.device)
        rstd = torch.empty(M, device=x.device)
        check(lib().genrl_ln_act_fwd(_p(x2), N, _p(gamma), _p(beta), _p(y), N, _p(mean), _p(rstd), M, N, eps, act,
                                     _stream()), 'ln_act_fwd')
        ctx.save_for_backward(x2, gamma, beta, mean, rstd)
        ctx.act = act
        ctx.xshape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, beta, mean, rstd = ctx.saved_tensors
        M, N = x2.shape
        dy2 = dy.reshape(M, N).contiguous()
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        gb = torch.empty(2, N, device=dy.device) if need_p else None
        dg, db = (gb[0], gb[1]) if need_p else (None, None)
        ws = _ws(lib().genrl_ln_ws_floats(M, N), dy.device) if need_p else None
        check(lib().genrl_ln_act_bwd(_p(dy2), N, _p(x2), N, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dx), N,
                                     _p(dg), _p(db), None, _p(ws), M, N, ctx.act, 0, _stream()), 'ln_act_bwd')
        return (dx.reshape(ctx.xshape) if dx is not None else None), dg, db, None, None


def ln_act(x, gamma, beta, eps=1e-5, act=True):
    return _LNAct.apply(x, gamma, beta, float(eps), int(act))


# ------------------------------------------------------------------ GRU gates

def _gru_fwd_raw(pre_ptr, h_ptr, gamma, beta, out_ptr, out2_ptr, scale2_ptr, mean_ptr, rstd_ptr, R, D):
    check(lib().genrl_gru_gates_fwd(pre_ptr, h_ptr, D, _p(gamma), _p(beta), out_ptr, D, out2_ptr, scale2_ptr, mean_ptr,
                                    rstd_ptr, R, D, 1e-5, _stream()), 'gru_gates_fwd')


def _gru_bwd_raw(dout_ptr, dout2_ptr, scale2_ptr, pre_ptr, h_ptr, gamma, beta, mean_ptr, rstd_ptr, dpre_ptr, dh_ptr,
                 dg, db, ws, R, D, accumulate, parts_ptr=None, nparts=0, part_stride=0):
    check(lib().genrl_gru_gates_bwd(dout_ptr, D, dout2_ptr, scale2_ptr, pre_ptr, h_ptr, D, _p(gamma), _p(beta),
                                    mean_ptr, rstd_ptr, dpre_ptr, dh_ptr, D, _p(dg), _p(db), _p(ws), R, D,
                                    int(accumulate), parts_ptr, nparts, part_stride, _stream()), 'gru_gates_bwd')


class _GRUGates(Function):
    @staticmethod
    def forward(ctx, pre, h, gamma, beta):
        pre = _f32(pre).contiguous(); h = _f32(h).contiguous()
        R, D = h.shape
        out = torch.empty_like(h)
        mean = torch.empty(R, device=h.device); rstd = torch.empty(R, device=h.device)
        _gru_fwd_raw(_p(pre), _p(h), gamma, beta, _p(out), None, None, _p(mean), _p(rstd), R, D)
        ctx.save_for_backward(pre, h, gamma, beta, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        pre, h, gamma, beta, mean, rstd = ctx.saved_tensors
        R, D = h.shape
        dout = dout.contiguous()
        dpre = torch.empty_like(pre); dh = torch.empty_like(h)
        need_p = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        gb = torch.empty(2, 3 * D, device=h.device) if need_p else None
        ws = _ws(lib().genrl_gru_ws_floats(R, D), h.device) if need_p else None
        _gru_bwd_raw(_p(dout), None, None, _p(pre), _p(h), gamma, beta, _p(mean), _p(rstd), _p(dpre), _p(dh),
                     gb[0] if need_p else None, gb[1] if need_p else None, ws, R, D, False)
        return dpre, dh, (gb[0] if need_p else None), (gb[1] if need_p else None)


def gru_gates(pre, h, gamma, beta):
    return _GRUGates.apply(pre, h, gamma, beta)


# ------------------------------------------------------------------ categorical latents

class _OneHotSample(Function):
    @staticmethod
    def forward(ctx, logits, q):
        lg = _f32(logits).contiguous()
        K = lg.shape[-1]
        G = lg.numel() // K
        out = torch.empty_like(lg)
        check(lib().genrl_onehot_fwd(_p(lg), _p(q.contiguous()), _p(out), None, G, K, UNIMIX, _stream()), 'onehot_fwd')
        ctx.save_for_backward(lg)
        return out

    @staticmethod
    def backward(ctx, g):
        (lg,) = ctx.saved_tensors
        K = lg.shape[-1]
        d = torch.empty_like(lg)
        check(lib().genrl_onehot_bwd(_p(lg), _p(g.contiguous()), _p(d), lg.numel() // K, K, UNIMIX, 0, _stream()),
              'onehot_bwd')
        return d, None


def onehot_sample(logits, q):
    """OneHotDist.sample with exponential noise q (same numel as logits); straight-through grad."""
    return _OneHotSample.apply(logits, q)


class _CatKL(Function):
    @staticmethod
    def forward(ctx, lp, lq):
        lp = _f32(lp).contiguous(); lq = _f32(lq).contiguous()
        S, K = lp.shape[-2:]
        R = lp.numel() // (S * K)
        kl = torch.empty(R, device=lp.device)
        check(lib().genrl_cat_kl_fwd(_p(lp), _p(lq), _p(kl), None, None, R, S, K, UNIMIX, _stream()), 'cat_kl_fwd')
        ctx.save_for_backward(lp, lq)
        return kl.reshape(lp.shape[:-2])

    @staticmethod
    def backward(ctx, g):
        lp, lq = ctx.saved_tensors
        S, K = lp.shape[-2:]
        R = lp.numel() // (S * K)
        g = g.reshape(R).contiguous()
        dlp = torch.empty_like(lp) if ctx.needs_input_grad[0] else None
        dlq = torch.empty_like(lq) if ctx.needs_input_grad[1] else None
        check(lib().genrl_cat_kl_bwd(_p(lp), _p(lq), _p(g), _p(g), _p(dlp), _p(dlq), R, S, K, UNIMIX, _stream()),
              'cat_kl_bwd')
        return dlp, dlq


def cat_kl(lp, lq):
    return _CatKL.apply(lp, lq)


def _time_major(x):
    """(B,T,..) views of contiguous (T,B,..) buffers (what EnsembleRSSM.observe hands out) are processed in their storage
    order -- the per-row results come back as the matching transposed view -- instead of being copied first."""
    if x.dim() >= 3 and not x.is_contiguous() and x.transpose(0, 1).is_contiguous():
        return x.transpose(0, 1), True
    return x.contiguous(), False


class _KLBalance(Function):
    """EnsembleRSSM.kl_loss's arithmetic as one node (agent/dreamer_utils.py:534-555, balance != 0.5, free_avg False):
    loss = mix * mean(max(KL(l || sg r), free)) + (1 - mix) * mean(max(KL(sg l || r), free)), plus the per-row KL.
    Forward: the KL kernel + a one-workgroup clamp-mean; backward: the per-row upstream gradients of both sides
    and ONE KL-backward launch (gp scales dl, gq scales dr) -- the elementwise formulation was ~25 launches."""
    @staticmethod
    def forward(ctx, l, r, mix, free):
        ctx.set_materialize_grads(False)         # (no zero tensor for the metric output's absent gradient)
        l, tl = _time_major(_f32(l)); r, tr = _time_major(_f32(r))
        if tl != tr:
            l, r = (l.transpose(0, 1).contiguous() if tl else l), (r.transpose(0, 1).contiguous() if tr else r)
            tl = tr = False
        ctx.tm = tl
        S, K = l.shape[-2:]
        R = l.numel() // (S * K)
        kl = torch.empty(R, device=l.device)
        check(lib().genrl_cat_kl_fwd(_p(l), _p(r), _p(kl), None, None, R, S, K, UNIMIX, _stream()), 'cat_kl_fwd')
        loss = torch.empty((), device=l.device)
        check(lib().genrl_kl_balance_fwd(_p(kl), R, mix, free, _p(loss), _stream()), 'kl_balance_fwd')
        ctx.save_for_backward(l, r, kl)
        ctx.mix, ctx.free = mix, free
        value = kl.reshape(l.shape[:-2])
        if tl:
            value = value.transpose(0, 1)
        ctx.mark_non_differentiable(value)
        return loss, value

    @staticmethod
    def backward(ctx, gloss, _gvalue):
        l, r, kl = ctx.saved_tensors
        S, K = l.shape[-2:]
        R = kl.numel()
        gp = torch.empty(R, device=l.device); gq = torch.empty(R, device=l.device)
        check(lib().genrl_kl_balance_bwd(_p(kl), _p(gloss.contiguous()), R, ctx.mix, ctx.free, _p(gp), _p(gq), _stream()),
              'kl_balance_bwd')
        dl = torch.empty_like(l) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(r) if ctx.needs_input_grad[1] else None
        check(lib().genrl_cat_kl_bwd(_p(l), _p(r), _p(gp), _p(gq), _p(dl), _p(dr), R, S, K, UNIMIX, _stream()), 'cat_kl_bwd')
        if ctx.tm:
            dl = dl.transpose(0, 1) if dl is not None else None
            dr = dr.transpose(0, 1) if dr is not None else None
        return dl, dr, None, None


def kl_balance(l, r, mix, free):
    """-> (scalar loss, per-row KL value [detached])"""
    return _KLBalance.apply(l, r, float(mix), float(free))


def cat_entropy(logits):
    lg, tm = _time_major(logits.detach())
    S, K = lg.shape[-2:]
    R = lg.numel() // (S * K)
    kl = torch.empty(R, device=lg.device); ent = torch.empty(R, device=lg.device)
    check(lib().genrl_cat_kl_fwd(_p(lg), _p(lg), _p(kl), _p(ent), None, R, S, K, UNIMIX, _stream()), 'cat_kl_fwd')
    ent = ent.reshape(lg.shape[:-2])
    return ent.transpose(0, 1) if tm else ent


# ------------------------------------------------------------------ two-hot

_buckets = {}


def twohot_buckets(dev):
    k = str(dev)
    if k not in _buckets:
        _buckets[k] = torch.linspace(-20.0, 20.0, steps=255, device=dev)
    return _buckets[k]


def _rows255(t):
    """(R x 255 view, row stride) of two-hot logits without copying when the rows are evenly spaced (the padded
    rows _Linear hands out for N % 4 != 0); otherwise a contiguous copy."""
    R = t.numel() // 255
    try:
        v = t.view(R, 255)
    except RuntimeError:
        v = None
    if v is None or v.stride(1) != 1 or (R > 1 and v.stride(0) < 255):
        v = t.reshape(R, 255).contiguous()
    return v, (v.stride(0) if R > 1 else 255)


class _TwoHot(Function):
    @staticmethod
    def forward(ctx, logits, x, mode):
        lg, ld = _rows255(_f32(logits))
        R = lg.shape[0]
        b = twohot_buckets(lg.device)
        xx = x.reshape(R).contiguous() if x is not None else None
        out = torch.empty(R, device=lg.device)
        check(lib().genrl_twohot_fwd(_prows(lg), ld, _p(xx), _p(b), _p(out), R, mode, _stream()), 'twohot_fwd')
        ctx.save_for_backward(lg, xx if xx is not None else lg.new_empty(0))
        ctx.mode, ctx.ld, ctx.lshape = mode, ld, logits.shape
        return out.reshape(logits.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        lg, xx = ctx.saved_tensors
        R = lg.shape[0]
        d = torch.empty(R, 256, device=lg.device)          # padded rows: the head's dgrad/wgrad read them in place
        check(lib().genrl_twohot_bwd(_prows(lg), ctx.ld, _p(xx) if ctx.mode == 0 else None, _p(twohot_buckets(lg.device)),
                                     _p(g.reshape(R).contiguous()), _p(d), 256, R, ctx.mode, _stream()), 'twohot_bwd')
        return d[:, :255].view(ctx.lshape), None, None


def twohot_logprob(logits, x):
    """TwoHotDist(logits).log_prob(x): logits (...,255), x (...,1) or (...) -> (...)"""
    return _TwoHot.apply(logits, x.detach(), 0)


def twohot_mean(logits):
    """TwoHotDist(logits).mean -> (...,1)"""
    return _TwoHot.apply(logits, None, 1).unsqueeze(-1)


# ------------------------------------------------------------------ lambda return

class _LambdaReturn(Function):
    @staticmethod
    def forward(ctx, reward, value, disc, lam):
        Hv = value.shape[0]
        H = Hv - 1
        N = value.numel() // Hv
        r = _f32(reward).contiguous(); v = _f32(value).contiguous()
        assert r.shape[0] in (H, H + 1) and r.numel() // r.shape[0] == N      # (H+1 rows: the last one is not read)
        out = torch.empty((H,) + tuple(r.shape[1:]), device=r.device)
        check(lib().genrl_lambda_return_fwd(_p(r), _p(v), _p(out), H, N, disc, lam, _stream()), 'lambda_fwd')
        ctx.dims = (H, N, disc, lam, value.shape, reward.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        H, N, disc, lam, vshape, rshape = ctx.dims
        g = g.contiguous()
        dr = torch.empty(rshape, device=g.device)
        dv = torch.empty(vshape, device=g.device)
        check(lib().genrl_lambda_return_bwd(_p(g), _p(dr), _p(dv), H, N, disc, lam, int(rshape[0] == H + 1), _stream()),
              'lambda_bwd')
        return dr, dv, None, None


def lambda_return(reward, value, disc, lam):
    """reward (H,N,1) -- or (H+1,N,1) with an unused last row: no slice, no slice-backward --, value (H+1,N,1)
    [value[-1] is the bootstrap] -> (H,N,1)"""
    return _LambdaReturn.apply(reward, value, float(disc), float(lam))


# ------------------------------------------------------------------ small statistics / scalar losses (stats.hip)

def moments(x):
    """-> tensor [mean, unbiased std, mean |x|, mean x^2] of all elements (one launch)"""
    x = _f32(x.detach()).contiguous()
    out = torch.empty(4, device=x.device)
    check(lib().genrl_moments(_p(x), x.numel(), _p(out), _stream()), 'moments')
    return out


def quantile_ema(x_flat, ema_vals, alpha, q0=0.05, q1=0.95):
    """RewardEMA in one launch: ema_vals updated in place; -> tensor [offset, scale, quantile0, quantile1]"""
    x = _f32(x_flat.detach()).contiguous()
    assert ema_vals.is_contiguous() and ema_vals.numel() == 2
    out = torch.empty(4, device=x.device)
    check(lib().genrl_quantile_ema(_p(x), x.numel(), q0, q1, alpha, _p(ema_vals), _p(out), _stream()), 'quantile_ema')
    return out


def normal_entropy_mean(raw, min_std, max_std):
    r2 = _f32(raw.detach()).reshape(-1, raw.shape[-1]).contiguous()
    out = torch.empty((), device=raw.device)
    ws = torch.empty(128, device=raw.device)
    check(lib().genrl_normal_entropy_mean(_p(r2), r2.shape[0], r2.shape[1] // 2, min_std, max_std, _p(out), _p(ws), _stream()),
          'normal_entropy_mean')
    return out


class _WMean(Function):
    @staticmethod
    def forward(ctx, x, w, scale):
        x = _f32(x).contiguous()
        w = _f32(w.detach()).expand_as(x).contiguous() if w is not None else None
        out = torch.empty((), device=x.device)
        check(lib().genrl_wmean_fwd(_p(x), _p(w), x.numel(), scale, _p(out), _stream()), 'wmean_fwd')
        ctx.w, ctx.scale, ctx.shape = w, scale, x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        dx = torch.empty(ctx.shape, device=g.device)
        check(lib().genrl_wmean_bwd(_p(g.contiguous()), _p(ctx.w), dx.numel(), ctx.scale, _p(dx), _stream()), 'wmean_bwd')
        return dx, None, None


class _CosDist(Function):
    @staticmethod
    def forward(ctx, x, c):
        x2 = _f32(x).reshape(-1, x.shape[-1]).contiguous(); c2 = _f32(c.detach()).reshape(-1, c.shape[-1]).contiguous()
        R, E = x2.shape
        cosv, xn = torch.empty(R, device=x.device), torch.empty(R, device=x.device)
        out = torch.empty((), device=x.device)
        check(lib().genrl_cosdist_fwd(_p(x2), _p(c2), _p(cosv), _p(xn), _p(out), R, E, _stream()), 'cosdist_fwd')
        ctx.save_for_backward(x2, c2, cosv, xn)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        x2, c2, cosv, xn = ctx.saved_tensors
        R, E = x2.shape
        dx = torch.empty_like(x2)
        check(lib().genrl_cosdist_bwd(_p(x2), _p(c2), _p(cosv), _p(xn), _p(g.contiguous()), _p(dx), R, E, _stream()), 'cosdist_bwd')
        return dx.reshape(ctx.shape), None


def cosine_distance(x, c):
    """1 - F.cosine_similarity(F.normalize(x, dim=-1), c, dim=-1).mean() as one node (gradient to x)"""
    return _CosDist.apply(x, c)


def connector_prep(video, eps, nf, lam, cscale):
    """-> clean (B,T,E), noisy (B,T,E), actions time-major (T,B,E+nf): VideoSSM.update's input preparation in one launch"""
    v = _f32(video).contiguous(); e = _f32(eps).contiguous()
    B, T, E = v.shape
    clean, noisy = torch.empty_like(v), torch.empty_like(v)
    act = torch.empty(T, B, E + nf, device=v.device)
    check(lib().genrl_connector_prep(_p(v), _p(e), _p(clean), _p(noisy), _p(act), B, T, E, nf, lam, cscale, _stream()),
          'connector_prep')
    return clean, noisy, act


def wmean(x, w=None, scale=1.0):
    """scale * mean(x * w) as ONE node (w: detached weights or None): the reference's `-(x * w).mean()` chains are
    mul + mean + neg forward and four elementwise launches backward."""
    if w is None:
        x, _ = _time_major(x)            # (a mean does not care about the order: transposed views are read in place)
    if not x.requires_grad:
        return _WMean.forward(_NoCtx(), x, w, float(scale))
    return _WMean.apply(x, w, float(scale))


class _NoCtx:
    pass


class _ActorObjective(Function):
    @staticmethod
    def forward(ctx, target, weight, os):
        t = _f32(target).contiguous()
        H = t.shape[0]
        N = t.numel() // H
        w = _f32(weight.detach()).reshape(-1).contiguous() if weight is not None else None
        assert w is None or w.numel() == (H - 1) * N
        loss, out = torch.empty((), device=t.device), torch.empty(2, device=t.device)
        check(lib().genrl_actor_obj_fwd(_p(t), _p(w), _p(os), H, N, _p(loss), _p(out), _stream()), 'actor_obj_fwd')
        ctx.save_for_backward(os)
        ctx.w, ctx.dims, ctx.shape = w, (H, N), t.shape
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, g, _g2):
        (os,) = ctx.saved_tensors
        H, N = ctx.dims
        d = torch.empty(ctx.shape, device=g.device)
        check(lib().genrl_actor_obj_bwd(_p(g.contiguous()), _p(ctx.w), _p(os), H, N, _p(d), _stream()), 'actor_obj_bwd')
        return d, None, None


def actor_objective(target, weight, offset_scale):
    """-> (loss, stats[2] = mean, std of the normalised returns): agent/dreamer.py:392-429 with actor_ent 0"""
    return _ActorObjective.apply(target, weight, offset_scale)


# ------------------------------------------------------------------ MSE image likelihood

class _MSELike(Function):
    @staticmethod
    def forward(ctx, mean, obs_u8):
        m = _f32(mean).contiguous(); o = obs_u8.contiguous()
        assert o.dtype == torch.uint8 and o.numel() == m.numel()
        N = m.shape[0]
        E = m.numel() // N
        like = torch.empty(N, device=m.device)
        check(lib().genrl_mse_fwd(_p(m), _p(o), _p(like), N, E, _stream()), 'mse_fwd')
        ctx.save_for_backward(m, o)
        return like

    @staticmethod
    def backward(ctx, g):
        m, o = ctx.saved_tensors
        N = m.shape[0]
        d = torch.empty_like(m)
        check(lib().genrl_mse_bwd(_p(m), _p(o), _p(g.contiguous()), _p(d), N, m.numel() // N, _stream()), 'mse_bwd')
        return d, None


def mse_like(mean, obs_u8):
    """MSEDist(mean).log_prob(obs/255-0.5) per frame: mean (N,C,H,W) f32, obs (N,C,H,W) u8 -> (N,)"""
    return _MSELike.apply(mean, obs_u8)


# ------------------------------------------------------------------ max-cosine reward

class _MaxCos(Function):
    @staticmethod
    def forward(ctx, u, v, urow):
        v2 = _f32(v).reshape(-1, v.shape[-1]).contiguous()
        u2 = _f32(u).reshape(-1, u.shape[-1]).contiguous()
        R, E = v2.shape
        ur = urow.reshape(R).contiguous() if urow is not None else None
        out = torch.empty(R, device=v.device)
        check(lib().genrl_maxcos_fwd(_p(u2), _p(v2), _p(ur), _p(out), R, E, _stream()), 'maxcos_fwd')
        ctx.save_for_backward(u2, v2, ur if ur is not None else torch.empty(0, device=v.device))
        ctx.has_idx = ur is not None
        ctx.vshape = v.shape
        return out.reshape(v.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        u2, v2, ur = ctx.saved_tensors
        R, E = v2.shape
        dv = torch.empty_like(v2)
        check(lib().genrl_maxcos_bwd(_p(u2), _p(v2), _p(ur) if ctx.has_idx else None, _p(g.reshape(R).contiguous()),
                                     _p(dv), R, E, _stream()), 'maxcos_bwd')
        return None, dv.reshape(ctx.vshape), None


def maxcos(u, v, urow=None):
    """max_cosine_similarity(u[urow], v) along the last dim; gradient flows to v only."""
    return _MaxCos.apply(u.detach(), v, urow)


# ------------------------------------------------------------------ actor head

class _ActorHead(Function):
    @staticmethod
    def forward(ctx, raw, eps, min_std, max_std):
        r2 = _f32(raw).reshape(-1, raw.shape[-1]).contiguous()
        R, A2 = r2.shape
        A = A2 // 2
        e = eps.reshape(R, A).contiguous()
        act = torch.empty(R, A, device=raw.device)
        check(lib().genrl_actor_head_fwd(_p(r2), _p(e), _p(act), None, None, R, A, min_std, max_std, 0, _stream()),
              'actor_head_fwd')
        ctx.save_for_backward(r2, e)
        ctx.cfg = (min_std, max_std, raw.shape)
        return act.reshape(*raw.shape[:-1], A)

    @staticmethod
    def backward(ctx, g):
        r2, e = ctx.saved_tensors
        min_std, max_std, rshape = ctx.cfg
        R, A2 = r2.shape
        d = torch.empty_like(r2)
        check(lib().genrl_actor_head_bwd(_p(g.reshape(R, A2 // 2).contiguous()), _p(r2), _p(e), _p(d), R, A2 // 2,
                                         min_std, max_std, 0, _stream()), 'actor_head_bwd')
        return d.reshape(rshape), None, None, None


def actor_sample(raw, eps, min_std=0.1, max_std=1.0):
    """raw (...,2A)=[out|std_raw] -> action = tanh(out) + std*eps (Normal rsample)"""
    return _ActorHead.apply(raw, eps, float(min_std), float(max_std))


def actor_mean_std(raw, min_std=0.1, max_std=1.0):
    r2 = raw.detach().reshape(-1, raw.shape[-1]).contiguous()
    R, A2 = r2.shape
    A = A2 // 2
    mean = torch.empty(R, A, device=raw.device); std = torch.empty(R, A, device=raw.device)
    check(lib().genrl_actor_head_fwd(_p(r2), None, None, _p(mean), _p(std), R, A, min_std, max_std, 0, _stream()),
          'actor_head_fwd')
    return mean.reshape(*raw.shape[:-1], A), std.reshape(*raw.shape[:-1], A)


# ---- genrl_rollout_f32 (include/genrl_hip.h): the arguments of the fp32-operand rollout's launch loops in C (csrc/seq.hip)
_FP, _CF, _CI = ctypes.c_void_p, ctypes.c_float, ctypes.c_int


class _RolloutF32Args(ctypes.Structure):
    _fields_ = ([(n, _CI) for n in ('H', 'N', 'S', 'K', 'D', 'A', 'AP', 'U', 'L')] + [(n, _CF) for n in ('unimix', 'min_std', 'max_std')]
                + [(n, _FP) for n in ('stoch', 'deter', 'logit', 'action', 'raws', 'eps', 'q', 'x_pre', 'x', 'g_pre', 'o_pre', 'o',
                                      'xm', 'xr', 'gm', 'gr', 'om', 'orr', 'ws_in', 'wa', 'gru_w', 'out_w', 'dist_w')]
                + [('in_b', _FP), ('in_g', _FP), ('in_be', _FP), ('in_eps', _CF), ('gru_g', _FP), ('gru_be', _FP),
                   ('out_b', _FP), ('out_g', _FP), ('out_be', _FP), ('out_eps', _CF), ('dist_b', _FP),
                   ('pw', _FP * 8), ('pb', _FP * 8), ('pg', _FP * 8), ('pbe', _FP * 8), ('peps', _CF * 8), ('pU', _CI * 8),
                   ('ppre', _FP * 8), ('py', _FP * 8), ('pmean', _FP * 8), ('prstd', _FP * 8), ('head_w', _FP), ('head_b', _FP),
                   ('ws', _FP), ('ws_floats', ctypes.c_long)]
                + [(n, _FP) for n in ('ds', 'dd', 'dl_in', 'dact_all', 'd_raw', 'dlg', 'dov', 'do_pre', 'dg_pre', 'dx', 'dx_pre', 'dha', 'dhb', 'waT')])


def _rollout_f32_args(sp, tape, dims, bufs, ws_in, wa, x=None, o=None, eps=None, q=None):
    """the fields both directions share; -> (args, the workspace tensor that must outlive the call)"""
    H, N, S, K, D, A, U, AP = dims
    stoch, deter, logit, action, raws, x_pre, g_pre, o_pre, st = bufs
    a = _RolloutF32Args()
    a.H, a.N, a.S, a.K, a.D, a.A, a.AP, a.U, a.L = H, N, S, K, D, A, AP, U, len(tape.layers)
    a.unimix, a.min_std, a.max_std = UNIMIX, sp.min_std, sp.max_std
    for n_, t_ in (('stoch', stoch), ('deter', deter), ('logit', logit), ('action', action), ('raws', raws), ('eps', eps), ('q', q), ('x_pre', x_pre),
                   ('x', x), ('g_pre', g_pre), ('o_pre', o_pre), ('o', o), ('xm', st['xm']), ('xr', st['xr']), ('gm', st['gm']), ('gr', st['gr']),
                   ('om', st['om']), ('orr', st['or']), ('ws_in', ws_in), ('wa', wa), ('gru_w', sp.gru_w), ('out_w', sp.out_w), ('dist_w', sp.dist_w),
                   ('in_b', sp.in_b), ('in_g', sp.in_g), ('in_be', sp.in_be), ('gru_g', sp.gru_g), ('gru_be', sp.gru_be), ('out_b', sp.out_b),
                   ('out_g', sp.out_g), ('out_be', sp.out_be), ('dist_b', sp.dist_b), ('head_w', tape.head_w), ('head_b', tape.head_b)):
        setattr(a, n_, _p(t_))
    a.in_eps, a.out_eps = sp.in_eps, sp.out_eps
    for l, (W_, b_, ga_, be_, eps_) in enumerate(tape.layers):
        a.pw[l], a.pb[l], a.pg[l], a.pbe[l], a.peps[l], a.pU[l] = _p(W_), _p(b_), _p(ga_), _p(be_), eps_, W_.shape[0]
        a.ppre[l], a.py[l], a.pmean[l], a.prstd[l] = _p(tape.pre[l]), _p(tape.y[l]), _p(tape.mean[l]), _p(tape.rstd[l])
    SK = S * K
    shapes = [(N, U, SK), (N, U, D), (N, U, U), (N, U, AP), (N, 3 * D, U), (N, 3 * D, D), (N, SK, U), (N, D, U), (N, D, 3 * D), (N, U, 3 * D)]
    shapes += [(N, l_[0].shape[0], l_[0].shape[0]) for l_ in tape.layers]
    nws = max(lib().genrl_sgemm_ws_floats(*s_) for s_ in shapes)
    ws = torch.empty(max(nws, 1), device=deter.device)
    a.ws, a.ws_floats = ws.data_ptr(), nws
    return a, ws


# ------------------------------------------------------------------ policy over an imagination rollout

class ActorTape:
    """Activations of the policy MLP over one imagination rollout, time-major (H, N, units).

    WorldModel.imagine (agent/dreamer.py:262-270) evaluates the actor once per step on sg(feat_t), so
    its forward is sequential — but its BACKWARD is not: the input is detached, hence the H backward
    passes depend only on d(raw_t), which the BPTT chain through the frozen dynamics delivers one per
    step.  Each step's autograd node just files d(raw_t) here; the node of step 0 (necessarily the last
    to run) then does ONE LayerNorm backward, ONE dgrad and ONE wgrad GEMM per layer over all H*N rows
    (K = H*N = 16384 at c2) instead of H small ones — the same sums, ~H x fewer launches."""
    def __init__(self, H, N, layers, head_w, head_b, dev):
        self.H, self.N = H, N
        self.layers = layers                      # [(W, b, gamma, beta, eps)]
        self.head_w, self.head_b = head_w, head_b
        U = [l[0].shape[0] for l in layers]
        self.pre = [torch.empty(H, N, u, device=dev) for u in U]
        self.y = [torch.empty(H, N, u, device=dev) for u in U]
        self.mean = [torch.empty(H, N, device=dev) for _ in U]
        self.rstd = [torch.empty(H, N, device=dev) for _ in U]
        self.d_raw = torch.zeros(H, N, head_w.shape[0], device=dev)
        self.inputs = None                        # (x1 (H,N,K1), x2 (H,N,K2)) set by the caller after the rollout
        self.seen = 0

    head_leaves = None      # (W_mean, b_mean, W_std, b_std): the leaf parameters head_w / head_b were stacked from (set by the caller)

    def head_param_grads(self, d, x_last):
        """gradients of the stacked output layer from d = d(raw) (M, 2A) and its input x_last (M, U): under the Optimizer they are
        accumulated straight into the flat gradient buffers of the LEAF parameters the stack was built from (the mean head's rows, then
        the std head's) and (None, None) is returned -- autograd then has nothing to un-stack and nothing to add; else (dW, db) of the stack"""
        M, A2 = d.shape
        U = x_last.shape[-1]
        dev = d.device
        if self.head_leaves is not None:
            bufs = [_grad_buf(p_) for p_ in self.head_leaves]
            if all(b_ is not None for b_ in bufs):
                A = A2 // 2
                if A % 4 == 0:          # (both halves of d start on 16-byte boundaries: two products straight into the buffers)
                    for i in range(2):
                        sgemm(d, 1, A2, x_last, 1, U, bufs[2 * i], U, None, A, U, M, accumulate=True, a_off=i * A)
                        colsum(d[:, i * A:(i + 1) * A], out=bufs[2 * i + 1], accumulate=True, ld=A2)
                    return None, None
                dWh = torch.empty(A2, U, device=dev)
                sgemm(d, 1, A2, x_last, 1, U, dWh, U, None, A2, U, M)
                dbh = colsum(d)
                for i in range(2):
                    copy2d(dWh, U, bufs[2 * i], U, A, U, None, True, src_off=i * A * U)
                    copy2d(dbh, A2, bufs[2 * i + 1], A, 1, A, None, True, src_off=i * A)
                return None, None
        tgt = _grad_buf(self.head_w)
        dWh = None if tgt is not None else torch.empty(A2, U, device=dev)
        sgemm(d, 1, A2, x_last, 1, U, tgt if tgt is not None else dWh, U, None, A2, U, M, accumulate=tgt is not None)
        tb = _grad_buf(self.head_b)
        dbh = None
        if tb is not None:
            colsum(d, out=tb, accumulate=True)
        else:
            dbh = colsum(d)
        return dWh, dbh

    def step(self, t, x1, x2):
        flat = [q for l in self.layers for q in l[:4]]
        return _ActorStep.apply(x1, x2, self.head_w, self.head_b, self, t, *flat)

    def _forward(self, t, x1, x2, out=None, head=True):
        N = self.N
        a, c = _f32(x1).contiguous(), _f32(x2).contiguous()
        K1, K2 = a.shape[1], c.shape[1]
        x, Kx = None, None
        for l, (W, b, gamma, beta, eps) in enumerate(self.layers):
            U, K = W.shape
            pre, y = self.pre[l], self.y[l]
            off = t * N * U
            if l == 0:
                sgemm(a, K1, 1, W, K, 1, pre, U, b, N, U, K1, c_off=off)
                sgemm(c, K2, 1, W, K, 1, pre, U, None, N, U, K2, accumulate=True, b_off=K1, c_off=off)
            else:
                sgemm(x, Kx, 1, W, K, 1, pre, U, b, N, U, Kx, a_off=t * N * Kx, c_off=off)
            check(lib().genrl_ln_act_fwd(pre.data_ptr() + 4 * off, U, _p(gamma), _p(beta), y.data_ptr() + 4 * off, U,
                                         self.mean[l].data_ptr() + 4 * t * N, self.rstd[l].data_ptr() + 4 * t * N,
                                         N, U, eps, 1, _stream()), 'ln_act_fwd')
            x, Kx = y, U
        if not head:                  # the caller runs the output layer fused with the Normal head (head_fused)
            return None
        A2 = self.head_w.shape[0]
        raw = out if out is not None else torch.empty(N, A2, device=a.device)     # (out: the rollout's own (N, 2A) row)
        sgemm(x, Kx, 1, self.head_w, Kx, 1, raw, A2, self.head_b, N, A2, Kx, a_off=t * N * Kx)
        return raw

    def head_fused(self, t, eps_ptr, raw_ptr, action_ptr, ld_action, min_std, max_std, P=None, row0=0):
        """raw_t = y_last W_head^T + b, mean / std / action = head(raw_t, eps) in ONE launch (genrl_actor_head_linear_fwd);
        P: Planes of the actions (rows row0 ..) to fill, or None"""
        N = self.N
        A2, U = self.head_w.shape
        y = self.y[-1]
        check(lib().genrl_actor_head_linear_fwd(y.data_ptr() + 4 * t * N * U, U, _p(self.head_w), _p(self.head_b), eps_ptr, raw_ptr,
                                                action_ptr, N, U, A2 // 2, min_std, max_std, ld_action,
                                                P.ptr(row0) if P is not None else None, P.ld if P is not None else 0,
                                                P.plane if P is not None else 0, P.inv_ptr(row0) if P is not None else None,
                                                _stream()), 'actor_head_linear_fwd')

    def _backward(self):
        assert self.inputs is not None, 'ActorTape.inputs (time-major rollout states) not set'
        H, N = self.H, self.N
        M = H * N
        dev = self.d_raw.device
        A2, U = self.head_w.shape
        d = self.d_raw.reshape(M, A2)
        x_last = self.y[-1]
        # parameter gradients go straight into the optimiser's flat gradient buffers (GEMM / reduction epilogues
        # with accumulate) when the parameters have them -- autograd then has nothing to add (one elementwise
        # launch per parameter otherwise); the returned gradient is None for those
        dWh, dbh = self.head_param_grads(d, x_last)
        dy = torch.empty(M, U, device=dev)
        sgemm(d, A2, 1, self.head_w, 1, U, dy, U, None, M, U, A2)
        grads = [None] * len(self.layers)
        for l in range(len(self.layers) - 1, -1, -1):
            W, b, gamma, beta, eps = self.layers[l]
            U, K = W.shape
            dpre = torch.empty(M, U, device=dev)
            tg, tbe, tc = _grad_buf(gamma), _grad_buf(beta), (_grad_buf(b) if b is not None else None)
            direct = tg is not None and tbe is not None and (b is None or tc is not None)
            if direct:
                g0, g1, g2, acc_p = tg, tbe, tc, 1
            else:
                gb = torch.empty(3, U, device=dev)
                g0, g1, g2, acc_p = gb[0], gb[1], gb[2], 0
            ws = _ws(lib().genrl_ln_ws_floats(M, U), dev)
            if direct:
                acc_p |= defer_reduce(M, U, ws, g0, g1, g2)
            check(lib().genrl_ln_act_bwd(_p(dy), U, _p(self.pre[l]), U, _p(gamma), _p(beta), _p(self.mean[l]),
                                         _p(self.rstd[l]), _p(dpre), U, _p(g0), _p(g1), _p(g2), _p(ws), M, U, 1, acc_p,
                                         _stream()), 'ln_act_bwd')
            tw = _grad_buf(W)
            acc = tw is not None
            dW = tw if acc else torch.empty(U, K, device=dev)
            if l > 0:
                x = self.y[l - 1]
                sgemm(dpre, 1, U, x, 1, K, dW, K, None, U, K, M, accumulate=acc)
                dy = torch.empty(M, K, device=dev)
                sgemm(dpre, U, 1, W, 1, K, dy, K, None, M, K, U)
            else:
                x1, x2 = self.inputs
                K1, K2 = x1.shape[-1], x2.shape[-1]
                assert x1.is_contiguous() and x2.is_contiguous() and x1.shape[0] >= H and K1 + K2 == K
                sgemm(dpre, 1, U, x1, 1, K1, dW, K, None, U, K1, M, accumulate=acc)
                sgemm(dpre, 1, U, x2, 1, K2, dW, K, None, U, K2, M, c_off=K1, accumulate=acc)
            grads[l] = (None if acc else dW, None if (direct or b is None) else g2, None if direct else g0,
                        None if direct else g1)
        return dWh, dbh, grads


class _ActorStep(Function):
    @staticmethod
    def forward(ctx, x1, x2, head_w, head_b, tape, t, *params):
        ctx.tape, ctx.t, ctx.nparams = tape, t, len(params)
        return tape._forward(t, x1, x2)

    @staticmethod
    def backward(ctx, d_raw):
        tape, t = ctx.tape, ctx.t
        tape.d_raw[t].copy_(d_raw)
        tape.seen += 1
        if t != 0:
            return (None,) * (6 + ctx.nparams)
        assert tape.seen == tape.H, f'policy steps with gradient: {tape.seen} of {tape.H}'
        dWh, dbh, grads = tape._backward()
        flat = [g for lg in grads for g in lg]
        return (None, None, dWh, dbh, None, None, *flat)


class RolloutSpec:
    """Frozen world-model weights + shapes of one imagination rollout (plain container handed to _Rollout)."""
    def __init__(self, tape, in_w, in_b, in_g, in_be, in_eps, gru_w, gru_g, gru_be, out_w, out_b, out_g, out_be, out_eps,
                 dist_w, dist_b, S, K, min_std, max_std):
        self.tape = tape
        self.in_w, self.in_b, self.in_g, self.in_be, self.in_eps = in_w, in_b, in_g, in_be, float(in_eps)
        self.gru_w, self.gru_g, self.gru_be = gru_w, gru_g, gru_be
        self.out_w, self.out_b, self.out_g, self.out_be, self.out_eps = out_w, out_b, out_g, out_be, float(out_eps)
        self.dist_w, self.dist_b = dist_w, dist_b
        self.S, self.K, self.min_std, self.max_std = S, K, float(min_std), float(max_std)


def _ln_fwd_raw(pre_ptr, gamma, beta, y_ptr, mean_ptr, rstd_ptr, M, N, eps):
    check(lib().genrl_ln_act_fwd(pre_ptr, N, _p(gamma), _p(beta), y_ptr, N, mean_ptr, rstd_ptr, M, N, eps, 1, _stream()),
          'ln_act_fwd')


def _ln_bwd_raw(dy_ptr, pre_ptr, gamma, beta, mean_ptr, rstd_ptr, dpre_ptr, M, N):
    check(lib().genrl_ln_act_bwd(dy_ptr, N, pre_ptr, N, _p(gamma), _p(beta), mean_ptr, rstd_ptr, dpre_ptr, N, None, None, None,
                                 None, M, N, 1, 0, _stream()), 'ln_act_bwd')


class _Rollout(Function):
    """WorldModel.imagine's H-step loop (agent/dreamer.py:262-270) as ONE autograd node.

    Per step: policy(sg(feat)) -> rsample -> img_step (img_in + GRU + img_out + dist) -> one-hot sample.
    The world model is frozen here, so the backward is a pure dgrad chain; doing it in one node lets
    every gradient sum happen inside a kernel (GEMM `accumulate` epilogues into the cloned time-major
    gradient buffers, the gate-backward's two-input add) instead of ~150 autograd add / stack / zero
    kernels per rollout, and hands the policy's gradients to ActorTape in place.
    Returns time-major stoch (H+1,N,S,K), deter (H+1,N,D), logit (H+1,N,S,K), action (H+1,N,A), raw (H,N,2A)."""
    @staticmethod
    def forward(ctx, stoch0, deter0, logit0, eps, q, spec, head_w, head_b, *actor_params):
        ctx.set_materialize_grads(False)
        sp, tape = spec, spec.tape
        H, N = tape.H, tape.N
        S, K = sp.S, sp.K
        SK, D = S * K, deter0.shape[1]
        A = eps.shape[-1]
        U = sp.in_w.shape[0]
        dev = deter0.device
        f = lambda *shape: torch.empty(*shape, device=dev)
        # actions live in rows of AP = A rounded up to 4 floats (zero padded) and the action slice of the input
        # layer's weight is copied once into a (U, AP) zero-padded matrix: both thin products of a step
        # (x += action W_a^T, d action = dx W_a) then meet the vector-load preconditions of the GEMM
        AP = (A + 3) // 4 * 4
        stoch = f(H + 1, N, SK); deter = f(H + 1, N, D); logit = f(H + 1, N, SK); action = torch.zeros(H + 1, N, AP, device=dev)
        wa = torch.zeros(sp.in_w.shape[0], AP, device=dev)
        wa[:, :A].copy_(sp.in_w[:, SK:SK + A])
        ws_in = sp.in_w[:, :SK].contiguous()     # stoch slice with 16-byte aligned rows (the (U, SK + A) weight's are not)
        raws = f(H, N, 2 * A)
        stoch[0].copy_(stoch0.reshape(N, SK)); deter[0].copy_(deter0); logit[0].copy_(logit0.reshape(N, SK))
        x_pre, x = f(H, N, U), f(H, N, U)
        g_pre = f(H, N, 3 * D)
        o_pre, o = f(H, N, U), f(H, N, U)
        st = {k: f(H, N) for k in ('xm', 'xr', 'gm', 'gr', 'om', 'or')}
        eps = _f32(eps).contiguous(); q = _f32(q).contiguous()
        Kin, Kg = sp.in_w.shape[1], sp.gru_w.shape[1]
        pt = lambda t, off: t.data_ptr() + 4 * off
        seq_c = SEQ_C and gemm_profile is None and len(tape.layers) <= 8
        if seq_c:
            # the H-step launch loop in C (csrc/seq.hip: genrl_imagine_seq_f32_fwd -- the loop below, launch for launch)
            a, ws_keep = _rollout_f32_args(sp, tape, (H, N, S, K, D, A, U, AP), (stoch, deter, logit, action, raws, x_pre, g_pre, o_pre, st),
                                           ws_in, wa, x=x, o=o, eps=eps, q=q)
            check(lib().genrl_imagine_seq_f32_fwd(ctypes.addressof(a), _stream()), 'imagine_seq_f32_fwd')
        for h in (() if seq_c else range(H)):
            sN, dN = h * N * SK, h * N * D
            tape._forward(h, stoch[h], deter[h], head=False)
            tape.head_fused(h, pt(eps, h * N * A), pt(raws, h * N * 2 * A), pt(action, (h + 1) * N * AP), AP, sp.min_std, sp.max_std)
            # img_in: [stoch_h | action_{h+1}] -> hidden, LN + SiLU
            sgemm(stoch, SK, 1, ws_in, SK, 1, x_pre, U, sp.in_b, N, U, SK, a_off=sN, c_off=h * N * U)
            sgemm(action, AP, 1, wa, AP, 1, x_pre, U, None, N, U, AP, accumulate=True, a_off=(h + 1) * N * AP, c_off=h * N * U)
            _ln_fwd_raw(pt(x_pre, h * N * U), sp.in_g, sp.in_be, pt(x, h * N * U), pt(st['xm'], h * N), pt(st['xr'], h * N),
                        N, U, sp.in_eps)
            # GRU
            sgemm(x, U, 1, sp.gru_w, Kg, 1, g_pre, 3 * D, None, N, 3 * D, U, a_off=h * N * U, c_off=h * N * 3 * D)
            sgemm(deter, D, 1, sp.gru_w, Kg, 1, g_pre, 3 * D, None, N, 3 * D, D, accumulate=True, a_off=dN, b_off=U,
                  c_off=h * N * 3 * D)
            _gru_fwd_raw(pt(g_pre, h * N * 3 * D), pt(deter, dN), sp.gru_g, sp.gru_be, pt(deter, dN + N * D), None, None,
                         pt(st['gm'], h * N), pt(st['gr'], h * N), N, D)
            # prior head: img_out (+LN+SiLU), dist
            sgemm(deter, D, 1, sp.out_w, D, 1, o_pre, U, sp.out_b, N, U, D, a_off=dN + N * D, c_off=h * N * U)
            _ln_fwd_raw(pt(o_pre, h * N * U), sp.out_g, sp.out_be, pt(o, h * N * U), pt(st['om'], h * N), pt(st['or'], h * N),
                        N, U, sp.out_eps)
            sgemm(o, U, 1, sp.dist_w, U, 1, logit, SK, sp.dist_b, N, SK, U, a_off=h * N * U, c_off=sN + N * SK)
            check(lib().genrl_onehot_fwd(pt(logit, sN + N * SK), pt(q, h * N * SK), pt(stoch, sN + N * SK), None, N * S, K,
                                         UNIMIX, _stream()), 'onehot_fwd')
        tape.inputs = (stoch, deter)
        ctx.sp = sp
        ctx.bufs = (stoch, deter, logit, raws, eps, x_pre, g_pre, o_pre, st, wa, ws_in)
        ctx.nparams = len(actor_params)
        ctx.dims = (H, N, S, K, D, A, U)
        # (views, never the buffers themselves: ctx.bufs holds `deter` and `raws`, and an OUTPUT tensor kept on ctx is a reference cycle
        # through its grad_fn that Python's collector cannot see -- every eager iteration's rollout buffers, 1.4 GiB, stayed alive)
        return stoch.reshape(H + 1, N, S, K), deter.view(H + 1, N, D), logit.reshape(H + 1, N, S, K), action[:, :, :A], raws.view(H, N, 2 * A)

    @staticmethod
    def backward(ctx, d_stoch, d_deter, d_logit, d_action, d_raws):
        sp, tape = ctx.sp, ctx.sp.tape
        stoch, deter, logit, raws, eps, x_pre, g_pre, o_pre, st, wa, ws_in = ctx.bufs
        H, N, S, K, D, A, U = ctx.dims
        SK = S * K
        dev = deter.device
        z = lambda *shape: torch.zeros(*shape, device=dev)
        ds = d_stoch.reshape(H + 1, N, SK).clone() if d_stoch is not None else z(H + 1, N, SK)
        dd = d_deter.clone() if d_deter is not None else z(H + 1, N, D)
        dl_in = d_logit.reshape(H + 1, N, SK).contiguous() if d_logit is not None else None
        da_in = d_action.contiguous() if d_action is not None else None
        Kin, Kg = sp.in_w.shape[1], sp.gru_w.shape[1]
        f = lambda *shape: torch.empty(*shape, device=dev)
        AP = wa.shape[1]
        dlg, do, do_pre, dg_pre, dx, dx_pre = f(N, SK), f(N, U), f(N, U), f(N, 3 * D), f(N, U), f(N, U)
        waT = wa[:, :A].t().contiguous()                  # (A, U): the action columns, for the fused head backward
        dha, dhb = f(N, D), f(N, D)
        cur, nxt = dha, None                              # ping-pong: recurrent gradient into deter_h from step h's GRU
        pt = lambda t, off: t.data_ptr() + 4 * off
        dact_all = None
        if da_in is not None:                 # upstream action gradients, once, in rows padded like the forward's actions
            dact_all = torch.zeros(H + 1, N, AP, device=dev)
            dact_all[:, :, :A].copy_(da_in)
        seq_c = SEQ_C and gemm_profile is None and len(tape.layers) <= 8
        if seq_c:
            a, ws_keep = _rollout_f32_args(sp, tape, (H, N, S, K, D, A, U, AP), (None, deter, logit, None, raws, x_pre, g_pre, o_pre, st),
                                           ws_in, wa, eps=eps)
            for n_, t_ in (('ds', ds), ('dd', dd), ('dl_in', dl_in), ('dact_all', dact_all), ('d_raw', tape.d_raw), ('dlg', dlg), ('dov', do),
                           ('do_pre', do_pre), ('dg_pre', dg_pre), ('dx', dx), ('dx_pre', dx_pre), ('dha', dha), ('dhb', dhb), ('waT', waT)):
                setattr(a, n_, _p(t_))
            check(lib().genrl_imagine_seq_f32_bwd(ctypes.addressof(a), _stream()), 'imagine_seq_f32_bwd')
        for h in (() if seq_c else range(H - 1, -1, -1)):
            sN, dN = h * N * SK, h * N * D
            # grad wrt stoch_{h+1} (complete in ds[h+1]) -> logits (straight-through), plus any direct logit gradient
            if dl_in is not None:
                dlg.copy_(dl_in[h + 1])
            check(lib().genrl_onehot_bwd(pt(logit, sN + N * SK), pt(ds, sN + N * SK), _p(dlg), N * S, K, UNIMIX,
                                         int(dl_in is not None), _stream()), 'onehot_bwd')
            sgemm(dlg, SK, 1, sp.dist_w, 1, U, do, U, None, N, U, SK)
            _ln_bwd_raw(_p(do), pt(o_pre, h * N * U), sp.out_g, sp.out_be, pt(st['om'], h * N), pt(st['or'], h * N), _p(do_pre),
                        N, U)
            sgemm(do_pre, U, 1, sp.out_w, 1, D, dd, D, None, N, D, U, accumulate=True, c_off=dN + N * D)
            # GRU: upstream = dd[h+1] (+ recurrent part from step h+1's GRU, held in `nxt`)
            _gru_bwd_raw(pt(dd, dN + N * D), nxt.data_ptr() if nxt is not None else None, None, pt(g_pre, h * N * 3 * D),
                         pt(deter, dN), sp.gru_g, sp.gru_be, pt(st['gm'], h * N), pt(st['gr'], h * N), _p(dg_pre), _p(cur),
                         None, None, None, N, D, False)
            sgemm(dg_pre, 3 * D, 1, sp.gru_w, 1, Kg, cur, D, None, N, D, 3 * D, accumulate=True, b_off=U)
            sgemm(dg_pre, 3 * D, 1, sp.gru_w, 1, Kg, dx, U, None, N, U, 3 * D)
            _ln_bwd_raw(_p(dx), pt(x_pre, h * N * U), sp.in_g, sp.in_be, pt(st['xm'], h * N), pt(st['xr'], h * N), _p(dx_pre), N, U)
            sgemm(dx_pre, U, 1, ws_in, 1, SK, ds, SK, None, N, SK, U, accumulate=True, c_off=sN)
            # d action_{h+1} = upstream (already sitting in its padded row of dact_all) + dx_pre W_a: accumulate epilogue
            check(lib().genrl_actor_head_linear_bwd(_p(dx_pre), U, _p(waT), pt(dact_all, (h + 1) * N * AP) if dact_all is not None else None,
                                                    AP, pt(raws, h * N * 2 * A), pt(eps, h * N * A), pt(tape.d_raw, h * N * 2 * A), N, U, A,
                                                    sp.min_std, sp.max_std, _stream()), 'actor_head_linear_bwd')
            nxt, cur = cur, (dhb if cur is dha else dha)
        if d_raws is not None:
            tape.d_raw += d_raws
        dWh, dbh, grads = tape._backward()
        flat = [g for lg in grads for g in lg]
        return (None, None, None, None, None, None, dWh, dbh, *flat)


def imagine_rollout(stoch0, deter0, logit0, eps, q, spec):
    tape = spec.tape
    flat = [qq for l in tape.layers for qq in l[:4]]
    return _Rollout.apply(stoch0, deter0, logit0, eps, q, spec, tape.head_w, tape.head_b, *flat)


# ------------------------------------------------------------------ stride-2 convolutions (NHWC)

def _im2col(x, Nimg, Hi, Wi, C, k, mode):
    Ho, Wo = (Hi - k) // 2 + 1, (Wi - k) // 2 + 1
    cols = torch.empty(Nimg * Ho * Wo, C * k * k, device=x.device)
    check(lib().genrl_im2col_s2(_p(x), _p(cols), Nimg, Hi, Wi, C, k, mode, _stream()), 'im2col')
    return cols


def _col2im(cols, bias, Nimg, Ha, Wa, C, k, Ho=0, Wo=0, nchw=False):
    ho = Ho if Ho > 0 else 2 * (Ha - 1) + k
    wo = Wo if Wo > 0 else 2 * (Wa - 1) + k
    shape = (Nimg, C, ho, wo) if nchw else (Nimg, ho, wo, C)
    out = torch.empty(shape, device=cols.device)
    check(lib().genrl_col2im_s2(_p(cols), _p(bias), _p(out), Nimg, Ha, Wa, C, k, Ho, Wo, int(nchw), _stream()), 'col2im')
    return out


def _ln_fwd_rows(pre2d, gamma, beta, eps):
    M, N = pre2d.shape
    y = torch.empty_like(pre2d)
    mean = torch.empty(M, device=pre2d.device); rstd = torch.empty(M, device=pre2d.device)
    check(lib().genrl_ln_act_fwd(_p(pre2d), N, _p(gamma), _p(beta), _p(y), N, _p(mean), _p(rstd), M, N, eps, 1,
                                 _stream()), 'ln_act_fwd')
    return y, mean, rstd


def _ln_bwd_rows(dy2d, pre2d, gamma, beta, mean, rstd, bias=None):
    """-> dpre, dgamma, dbeta, column sums of dpre (= the producing layer's bias gradient), one pass.  When gamma,
    beta (and the layer's `bias`) have flat gradient buffers the three sums are accumulated straight into them and
    returned as None (nothing left for autograd to add)."""
    M, N = pre2d.shape
    dpre = torch.empty_like(pre2d)
    tg, tb, tc = _grad_buf(gamma), _grad_buf(beta), _grad_buf(bias)
    direct = tg is not None and tb is not None and tc is not None
    if direct:
        g0, g1, g2 = tg, tb, tc
    else:
        gb = torch.empty(3, N, device=pre2d.device)
        g0, g1, g2 = gb[0], gb[1], gb[2]
    ws = _ws(lib().genrl_ln_ws_floats(M, N), pre2d.device)
    acc_p = int(direct) | (defer_reduce(M, N, ws, g0, g1, g2) if direct else 0)
    check(lib().genrl_ln_act_bwd(_p(dy2d), N, _p(pre2d), N, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dpre), N,
                                 _p(g0), _p(g1), _p(g2), _p(ws), M, N, 1, acc_p, _stream()), 'ln_act_bwd')
    return (dpre, None, None, None) if direct else (dpre, g0, g1, g2)


def _implicit_conv(img, C):
    """The GEMM can gather stride-2 patches itself when every patch segment is 16-byte addressable."""
    return img.dtype == torch.float32 and C % 4 == 0 and img.data_ptr() % 16 == 0


class _Conv2dS2(Function):
    """nn.Conv2d(k, stride 2) as patch-gather + GEMM.  x: f32 NHWC (N,H,W,C) or u8 NCHW (N,C,H,W)
    [preprocess fused]; Wp (Co, k*k*Ci) = weight permuted to (co, kh, kw, ci); returns NHWC."""
    @staticmethod
    def forward(ctx, x, Wp, b, k, gamma=None, beta=None, eps=0.0):
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
        if _implicit_conv(x, C):       # patches gathered by the GEMM's A loader straight from the image
            sgemm_conv(x, K, 1, Wp, K, 1, y, Co, b, M, Co, K, 1, (Hi, Wi, C, k))
        else:
            cols = _im2col(x, Nimg, Hi, Wi, C, k, 2 if u8 else 0)
            sgemm(cols, K, 1, Wp, K, 1, y, Co, b, M, Co, K)
        ctx.dims = (Nimg, Hi, Wi, C, k, u8)
        ctx.fused_ln = gamma is not None
        ctx.bias = b
        if ctx.fused_ln:           # channel-LayerNorm + SiLU of the layer (ImgChLayerNorm + act)
            out, mean, rstd = _ln_fwd_rows(y, gamma, beta, eps)
            ctx.save_for_backward(x, Wp, y, mean, rstd, gamma, beta)
            return out.reshape(Nimg, Ho, Wo, Co)
        ctx.save_for_backward(x, Wp)
        return y.reshape(Nimg, Ho, Wo, Co)

    @staticmethod
    def backward(ctx, dy):
        x, Wp = ctx.saved_tensors[:2]
        Nimg, Hi, Wi, C, k, u8 = ctx.dims
        Co = Wp.shape[0]
        K = C * k * k
        Ho, Wo = (Hi - k) // 2 + 1, (Wi - k) // 2 + 1
        M = Nimg * Ho * Wo
        dy2 = dy.reshape(M, Co).contiguous()
        dx = dW = db = dg = dbe = None
        if ctx.fused_ln:
            pre, mean, rstd, gamma, beta = ctx.saved_tensors[2:]
            dy2, dg, dbe, db_ln = _ln_bwd_rows(dy2, pre, gamma, beta, mean, rstd, ctx.bias)
        if ctx.needs_input_grad[1]:
            dW = torch.empty(Co, K, device=dy.device)
            if _implicit_conv(x, C) and Co % 4 == 0:
                sgemm_conv(dy2, 1, Co, x, 1, K, dW, K, None, Co, K, M, 2, (Hi, Wi, C, k))
            else:
                cols = _im2col(x, Nimg, Hi, Wi, C, k, 2 if u8 else 0)     # recomputed, not stored
                sgemm(dy2, 1, Co, cols, 1, K, dW, K, None, Co, K, M)
                del cols
        if ctx.needs_input_grad[2]:
            db = db_ln if ctx.fused_ln else colsum(dy2)
        if (not u8) and ctx.needs_input_grad[0]:
            dcols = torch.empty(M, K, device=dy.device)
            sgemm(dy2, Co, 1, Wp, 1, K, dcols, K, None, M, K, Co)     # dcols = dy W
            dx = _col2im(dcols, None, Nimg, Ho, Wo, C, k, Hi, Wi)
        return dx, dW, db, None, dg, dbe, None


def permuted(W):
    """the (A, k*k, B) permutation of the conv weight W (A, B, k, k), cached per optimiser step (planes.derived)"""
    from . import planes
    A, B, k, _ = W.shape
    return planes.derived(W, 'perm', lambda: transpose_last2_raw(W.detach().reshape(A, B, k * k)))


class _PermuteWeight(Function):
    """(A, B, k, k) conv weight -> (A, k*k, B) for the NHWC products; the gradient is permuted back straight INTO the
    parameter's flat gradient buffer when it has one (no tensor for autograd's AccumulateGrad to add)."""
    @staticmethod
    def forward(ctx, W):
        A, B, k, _ = W.shape
        ctx.W = W
        # (cached per optimiser step: the encoder / decoder weights are permuted once, not once per call -- planes.derived)
        return permuted(W).detach()

    @staticmethod
    def backward(ctx, g):
        W = ctx.W
        A, B, k, _ = W.shape
        tgt = _grad_buf(W)
        if tgt is not None:
            transpose_last2_raw(g.contiguous(), out=tgt.view(A, B, k * k), accumulate=True)
            return None
        return transpose_last2_raw(g.contiguous()).reshape(W.shape)


def conv2d_s2(x, W, b, ln=None, fp32_out=True, planes_out=True):
    """W (Co,Ci,k,k) in the reference layout; permuted per call to (Co, kh*kw*Ci) (gradient flows back
    through the permute).  ln = (gamma, beta, eps): channel-LayerNorm + SiLU fused into the same node."""
    Co, Ci, k, _ = W.shape
    Wp = _PermuteWeight.apply(W).reshape(Co, k * k * Ci)
    if ln is None:
        return _Conv2dS2.apply(x, Wp, b, k)
    return _Conv2dS2.apply(x, Wp, b, k, ln[0], ln[1], float(ln[2]))


CONVT_DIRECT = os.environ.get('GENRL_CONVT_DIRECT', '1') != '0'
# the same layer's backward with the patch operands gathered from dy itself (genrl_convt_small_co_bwd): dgrad 175 + wgrad 185 + reduce 23 us
# against im2col 180 + 147 + 133 + 32 us (the first version, with 4-byte gather loads at an 8-byte lane stride, took 358 + 202 + 62)
CONVT_DIRECT_BWD = os.environ.get('GENRL_CONVT_DIRECT_BWD', '1') != '0'


def _nchw_bias_grad(dy, bias):
    """bias gradient of an NCHW output (the decoder's frames): per-channel sum; under the Optimizer it is added straight into the
    parameter's flat gradient buffer (-> None: autograd's AccumulateGrad then has nothing to do for it)"""
    db = dy.sum((0, 2, 3))
    tgt = _grad_buf(bias)
    if tgt is None:
        return db
    copy2d(db, db.numel(), tgt, db.numel(), 1, db.numel(), None, True)
    return None


class _ConvT2dS2(Function):
    """nn.ConvTranspose2d(k, stride 2) as GEMM + gather-form col2im.  x NHWC (N,Hi,Wi,Ci);
    Wp (Ci, k*k*Co) = weight permuted to (ci, kh, kw, co); returns NHWC."""
    @staticmethod
    def forward(ctx, x, Wp, b, k, gamma=None, beta=None, eps=0.0, out_nchw=False):
        x = _f32(x).contiguous()
        Nimg, Hi, Wi, Ci = x.shape
        Nw = Wp.shape[1]
        Co = Nw // (k * k)
        M = Nimg * Hi * Wi
        assert not (out_nchw and gamma is not None)
        if Co <= 4 and k == 6 and Ci == 48 and CONVT_DIRECT and Wp.is_contiguous() and not _p16():
            # the 3-channel end of the decoder: gather form on the fp32 matrix cores, no cols matrix (genrl_convt_small_co_fwd)
            Ho, Wo = 2 * (Hi - 1) + k, 2 * (Wi - 1) + k
            y = torch.empty((Nimg, Co, Ho, Wo) if out_nchw else (Nimg, Ho, Wo, Co), device=x.device)
            if gemm_profile is not None:
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
            check(lib().genrl_convt_small_co_fwd(_p(x), _p(Wp), _p(b), _p(y), Nimg, Hi, Wi, Ci, Co, k, int(out_nchw), _stream()),
                  'convt_small_co_fwd')
            if gemm_profile is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                gemm_profile.append((Nimg * (Hi + 2) * (Wi + 2), 4 * Co, 9 * Ci, e0, e1, 'kk/convt_direct'))
        else:
            cols = torch.empty(M, Nw, device=x.device)
            sgemm(x, Ci, 1, Wp, 1, Nw, cols, Nw, None, M, Nw, Ci)         # cols = x W
            y = _col2im(cols, b, Nimg, Hi, Wi, Co, k, nchw=out_nchw)       # (the last decoder layer hands out NCHW frames)
        ctx.out_nchw = out_nchw
        ctx.dims = (Nimg, Hi, Wi, Ci, Co, k)
        ctx.fused_ln = gamma is not None
        ctx.bias = b
        if ctx.fused_ln:
            out, mean, rstd = _ln_fwd_rows(y.reshape(-1, Co), gamma, beta, eps)
            ctx.save_for_backward(x, Wp, y, mean, rstd, gamma, beta)
            return out.reshape(y.shape)
        ctx.save_for_backward(x, Wp)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Wp = ctx.saved_tensors[:2]
        Nimg, Hi, Wi, Ci, Co, k = ctx.dims
        Ho, Wo = 2 * (Hi - 1) + k, 2 * (Wi - 1) + k
        M, Nw = Nimg * Hi * Wi, Co * k * k
        dy = dy.contiguous()
        dg = dbe = db_ln = None
        if ctx.fused_ln:
            pre, mean, rstd, gamma, beta = ctx.saved_tensors[2:]
            dy, dg, dbe, db_ln = _ln_bwd_rows(dy.reshape(-1, Co), pre.reshape(-1, Co), gamma, beta, mean, rstd, ctx.bias)
            dy = dy.reshape(Nimg, Ho, Wo, Co)
        implicit = _implicit_conv(dy, Co) and Ci % 4 == 0 and not ctx.out_nchw
        dx = dW = db = None
        if (CONVT_DIRECT_BWD and ctx.out_nchw and not ctx.fused_ln and Co == 3 and k == 6 and Ci == 48 and Wp.is_contiguous()
                and not _p16()):
            # the decoder's 3-channel end: both gradients gather their patch operands from dy itself (genrl_convt_small_co_bwd), no im2col
            if ctx.needs_input_grad[0]:
                dx = torch.empty(Nimg, Hi, Wi, Ci, device=dy.device)
            ws = None
            if ctx.needs_input_grad[1]:
                dW = torch.empty(Ci, Nw, device=dy.device)
                ws = torch.empty(lib().genrl_convt_small_co_bwd_ws_floats(Ci, Co), device=dy.device)
            if gemm_profile is not None:
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
            check(lib().genrl_convt_small_co_bwd(_p(x), _p(Wp), _p(dy), _p(dx), _p(dW), _p(ws), Nimg, Hi, Wi, Ci, Co, k, _stream()),
                  'convt_small_co_bwd')
            if gemm_profile is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                gemm_profile.append((M, Ci, Nw, e0, e1, 'kk/convt_direct_bwd'))
            if ctx.needs_input_grad[2]:
                db = _nchw_bias_grad(dy, ctx.bias)
            return dx, dW, db, None, None, None, None, None
        dcols = None if implicit else _im2col(dy, Nimg, Ho, Wo, Co, k, 1 if ctx.out_nchw else 0)   # (M, Nw) patch matrix of dy
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, Ci, device=dy.device)
            if implicit:
                sgemm_conv(dy, Nw, 1, Wp, Nw, 1, dx, Ci, None, M, Ci, Nw, 1, (Ho, Wo, Co, k))
            else:
                sgemm(dcols, Nw, 1, Wp, Nw, 1, dx, Ci, None, M, Ci, Nw)   # dx = dcols W^T
            dx = dx.reshape(Nimg, Hi, Wi, Ci)
        if ctx.needs_input_grad[1]:
            dW = torch.empty(Ci, Nw, device=dy.device)
            if implicit:
                sgemm_conv(x, 1, Ci, dy, 1, Nw, dW, Nw, None, Ci, Nw, M, 2, (Ho, Wo, Co, k))
            else:
                sgemm(x, 1, Ci, dcols, 1, Nw, dW, Nw, None, Ci, Nw, M)    # dW = x^T dcols
        if ctx.needs_input_grad[2]:
            if ctx.fused_ln:
                db = db_ln
            elif ctx.out_nchw:
                db = _nchw_bias_grad(dy, ctx.bias)
            else:
                db = colsum(dy.reshape(-1, Co))
        return dx, dW, db, None, dg, dbe, None, None


def convT2d_s2(x, W, b, ln=None, out_nchw=False, fp32_out=True, planes_out=True):
    """W (Ci,Co,k,k) in the reference layout; permuted per call to (Ci, kh*kw*Co).  out_nchw: (N,Co,Ho,Wo) output
    (written that way by the overlap-add kernel; no LayerNorm fusion then)."""
    Ci, Co, k, _ = W.shape
    Wp = _PermuteWeight.apply(W).reshape(Ci, k * k * Co)
    if ln is None:
        return _ConvT2dS2.apply(x, Wp, b, k, None, None, 0.0, out_nchw)
    return _ConvT2dS2.apply(x, Wp, b, k, ln[0], ln[1], float(ln[2]))


class _TransposeLast2(Function):
    @staticmethod
    def forward(ctx, x):
        return transpose_last2_raw(x)

    @staticmethod
    def backward(ctx, g):
        return transpose_last2_raw(g.contiguous())


def transpose_last2(x):
    """(B,P,C) -> (B,C,P) contiguous"""
    return _TransposeLast2.apply(x)


# ------------------------------------------------------------------ optimiser

def grad_norm(g_flat, out, scale=1.0, step_inc=None):
    ws = _ws(lib().genrl_sqnorm_ws_floats(g_flat.numel()), g_flat.device)
    check(lib().genrl_grad_norm(_p(g_flat), g_flat.numel(), _p(out), _p(ws), scale, _p(step_inc), _stream()), 'grad_norm')
    return out


def adam_step(p, g, m, v, norm, gscale, clip, lr, eps, wd, step, b1=0.9, b2=0.999, step_dev=None, zero_grad=False):
    check(lib().genrl_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(norm), gscale, clip, lr, b1, b2, eps, wd,
                                step, _p(step_dev), int(zero_grad), _stream()), 'adam_step')


def scale_(p, s):
    check(lib().genrl_scale(_p(p), p.numel(), s, _stream()), 'scale')


# ------------------------------------------------------------------ fused multi-input layers

def _aligned_block(W, K1, M):
    """First column block W[:, :K1] of a two-input layer's weight for the GEMMs: (tensor, row spacing).  Rows of W
    are K floats apart; with K % 4 != 0 (1024 latent + 10 action inputs) they are not 16-byte aligned and the big
    block would take the scalar-load kernel, so it gets a compact copy (4 MB, once per call)."""
    K = W.shape[1]
    if K % 4 and K1 % 4 == 0 and K1 < K and M > 32:
        return W[:, :K1].contiguous(), K1
    return W, K


class _Linear2(Function):
    """y = [x1, x2] W^T + b without materialising the concatenation: W = [W1 | W2] column blocks
    (torch.cat([stoch, action]) -> _img_in, agent/dreamer_utils.py:461-462; feat -> MLP dense0)."""
    @staticmethod
    def forward(ctx, x1, x2, W, b):
        a = _f32(x1).reshape(-1, x1.shape[-1]).contiguous()
        c = _f32(x2).reshape(-1, x2.shape[-1]).contiguous()
        M, K1 = a.shape
        K2 = c.shape[1]
        N, K = W.shape
        assert K == K1 + K2 and c.shape[0] == M
        y = torch.empty(M, N, device=a.device)
        w1, ld1 = _aligned_block(W, K1, M)
        sgemm(a, K1, 1, w1, ld1, 1, y, N, b, M, N, K1)
        sgemm(c, K2, 1, W, K, 1, y, N, None, M, N, K2, accumulate=True, b_off=K1)
        ctx.w1 = (w1, ld1)
        ctx.save_for_backward(a, c, W)
        ctx.has_bias = b is not None
        ctx.shapes = (x1.shape, x2.shape)
        return y.reshape(*x1.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        a, c, W = ctx.saved_tensors
        M, K1 = a.shape
        K2 = c.shape[1]
        N, K = W.shape
        dy2 = dy.reshape(M, N).contiguous()
        d1 = d2 = dW = db = None
        if ctx.needs_input_grad[0]:
            d1 = torch.empty(M, K1, device=dy.device)
            sgemm(dy2, N, 1, ctx.w1[0], 1, ctx.w1[1], d1, K1, None, M, K1, N)
            d1 = d1.reshape(ctx.shapes[0])
        if ctx.needs_input_grad[1]:
            d2 = torch.empty(M, K2, device=dy.device)
            sgemm(dy2, N, 1, W, 1, K, d2, K2, None, M, K2, N, b_off=K1)
            d2 = d2.reshape(ctx.shapes[1])
        if ctx.needs_input_grad[2]:
            dW = torch.empty(N, K, device=dy.device)
            sgemm(dy2, 1, N, a, 1, K1, dW, K, None, N, K1, M)
            sgemm(dy2, 1, N, c, 1, K2, dW, K, None, N, K2, M, c_off=K1)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = colsum(dy2)
        return d1, d2, dW, db


def linear2(x1, x2, W, b=None):
    return _Linear2.apply(x1, x2, W, b)


class _GRUStep(Function):
    """One GRUCell step h' = GRU(x, h) (agent/dreamer_utils.py:771-785): W = [Wx | Wh] (3D, I+D),
    no bias, LayerNorm over 3D, gates fused.  Rows = imagination rows."""
    @staticmethod
    def forward(ctx, x, h, W, gamma, beta):
        x = _f32(x).contiguous(); h = _f32(h).contiguous()
        R, I = x.shape
        D = h.shape[1]
        K = I + D
        pre = torch.empty(R, 3 * D, device=x.device)
        sgemm(x, I, 1, W, K, 1, pre, 3 * D, None, R, 3 * D, I)
        sgemm(h, D, 1, W, K, 1, pre, 3 * D, None, R, 3 * D, D, accumulate=True, b_off=I)
        out = torch.empty_like(h)
        mean = torch.empty(R, device=x.device); rstd = torch.empty(R, device=x.device)
        _gru_fwd_raw(_p(pre), _p(h), gamma, beta, _p(out), None, None, _p(mean), _p(rstd), R, D)
        ctx.save_for_backward(x, h, W, gamma, beta, pre, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h, W, gamma, beta, pre, mean, rstd = ctx.saved_tensors
        R, I = x.shape
        D = h.shape[1]
        K = I + D
        dout = dout.contiguous()
        dpre = torch.empty_like(pre); dh = torch.empty_like(h)
        need_p = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        tg, tb = _grad_buf(gamma), _grad_buf(beta)
        direct = need_p and tg is not None and tb is not None
        gb = torch.empty(2, 3 * D, device=h.device) if (need_p and not direct) else None
        g0, g1 = (tg, tb) if direct else ((gb[0], gb[1]) if need_p else (None, None))
        ws = _ws(lib().genrl_gru_ws_floats(R, D), h.device) if need_p else None
        _gru_bwd_raw(_p(dout), None, None, _p(pre), _p(h), gamma, beta, _p(mean), _p(rstd), _p(dpre), _p(dh),
                     g0, g1, ws, R, D, direct)
        dx = dW = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            sgemm(dpre, 3 * D, 1, W, 1, K, dx, I, None, R, I, 3 * D)
        if ctx.needs_input_grad[1]:
            sgemm(dpre, 3 * D, 1, W, 1, K, dh, D, None, R, D, 3 * D, accumulate=True, b_off=I)
        else:
            dh = None
        if ctx.needs_input_grad[2]:
            tgt = _grad_buf(W)
            acc = tgt is not None
            if not acc:
                dW = tgt = torch.empty(3 * D, K, device=h.device)
            def wg():
                sgemm(dpre, 1, 3 * D, x, 1, I, tgt, K, None, 3 * D, I, R, accumulate=acc)
                sgemm(dpre, 1, 3 * D, h, 1, D, tgt, K, None, 3 * D, D, R, accumulate=acc, c_off=I)
            if acc:
                wgrad_stream.run(wg, dpre, x, h)
            else:
                wg()
        if need_p and not direct:
            return dx, dh, dW, g0, g1
        return dx, dh, dW, None, None


def gru_step(x, h, W, gamma, beta):
    return _GRUStep.apply(x, h, W, gamma, beta)


SEQ_C = os.environ.get('GENRL_SEQ_C', '1') != '0'      # the scans' per-step launch loops run in C (csrc/seq.hip); 0: from Python


class _GRUSeq(Function):
    """The whole GRU recurrence of EnsembleRSSM.observe / VideoSSM.update over T steps with the
    non-recurrent half hoisted: pre_x = x W_x^T for all T at once; per step only
    h_{t-1} W_h^T (+ LayerNorm + gates) is sequential: two launches per step.  x (T,B,I); mask (T,B)
    multiplies h_{t-1} (is_first reset, agent/dreamer_utils.py:433-434) or None; h0 (B,D).
    Returns deter (T,B,D)."""
    @staticmethod
    def forward(ctx, x, mask, h0, W, gamma, beta):
        x = _f32(x).contiguous()
        T, B, I = x.shape
        D = h0.shape[1]
        K = I + D
        dev = x.device
        pre = torch.empty(T, B, 3 * D, device=dev)
        sgemm(x, I, 1, W, K, 1, pre, 3 * D, None, T * B, 3 * D, I)
        out = torch.empty(T, B, D, device=dev)
        mean = torch.empty(T, B, device=dev); rstd = torch.empty(T, B, device=dev)
        h0 = _f32(h0).contiguous()
        if mask is not None:
            mask = mask.contiguous()
            hm = torch.empty(T, B, D, device=dev)       # masked previous state of every step
            copy2d(h0, D, hm, D, B, D, mask[0])
        else:
            hm = None
        BD, B3D = B * D, B * 3 * D
        seq_c = SEQ_C and gemm_profile is None
        if seq_c:
            # the T-step launch loop in C (csrc/seq.hip: same launches, same order -- one host call instead of 2 T)
            nws = lib().genrl_gru_seq_ws_floats(B, D)
            ws = torch.empty(nws, device=dev) if nws > 0 else None
            check(lib().genrl_gru_seq_fwd(_p(pre), W.data_ptr() + 4 * I, K, _p(gamma), _p(beta), _p(h0), _p(mask), _p(out), _p(hm), _p(mean),
                                          _p(rstd), _p(ws), nws, T, B, D, 1e-5, _stream()), 'gru_seq_fwd')
        for t in (range(T) if not seq_c else ()):
            if hm is not None:
                hprev, hoff = hm, t * BD
            else:
                hprev, hoff = (h0, 0) if t == 0 else (out, (t - 1) * BD)
            sgemm(hprev, D, 1, W, K, 1, pre, 3 * D, None, B, 3 * D, D, accumulate=True, a_off=hoff, b_off=I,
                  c_off=t * B3D)
            nxt = hm is not None and t + 1 < T
            _gru_fwd_raw(pre.data_ptr() + 4 * t * B3D, hprev.data_ptr() + 4 * hoff, gamma, beta,
                         out.data_ptr() + 4 * t * BD, (hm.data_ptr() + 4 * (t + 1) * BD) if nxt else None,
                         (mask.data_ptr() + 4 * (t + 1) * B) if nxt else None,
                         mean.data_ptr() + 4 * t * B, rstd.data_ptr() + 4 * t * B, B, D)
        ctx.save_for_backward(x, mask if mask is not None else x.new_empty(0), h0, W, gamma, beta, pre, out,
                              hm if hm is not None else x.new_empty(0), mean, rstd)
        ctx.has_mask = mask is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, mask, h0, W, gamma, beta, pre, out, hm, mean, rstd = ctx.saved_tensors
        T, B, I = x.shape
        D = out.shape[2]
        K = I + D
        dev = x.device
        dout = dout.contiguous()
        dpre = torch.empty_like(pre)
        dha = torch.empty(B, D, device=dev); dhb = torch.empty(B, D, device=dev)   # ping-pong d(hm_t)
        # the LayerNorm parameter gradients pile up per workgroup in `ws` over the scan (accumulate_params bits
        # 2|4) and are reduced once, by the last step (t = 0) -- straight into the flat gradient buffers when
        # gamma / beta have them
        tg, tb = _grad_buf(gamma), _grad_buf(beta)
        direct = tg is not None and tb is not None
        gb = (tg, tb) if direct else torch.empty(2, 3 * D, device=dev)
        ws = torch.empty(lib().genrl_gru_ws_floats(B, D), device=dev)
        BD, B3D = B * D, B * 3 * D
        # few sequences per GPU: the recurrent dgrad d(hm_t) += dpre_t W_h is a weight stream with only D/16
        # column blocks -> K-split into slabs that the next step's gate backward sums (no reduce launch)
        S = 4 if (B <= 32 and not _p16()) else 0
        pa = torch.empty(S, B, D, device=dev) if S else None
        pb = torch.empty(S, B, D, device=dev) if S else None
        cur, nxt, pcur, pnxt = dha, None, pa, None
        seq_c = SEQ_C and gemm_profile is None
        if seq_c:
            nws = lib().genrl_gru_seq_ws_floats(B, D)
            ws2 = torch.empty(nws, device=dev) if nws > 0 else None
            fin = (ctypes.c_int * 2)(0, -1)
            check(lib().genrl_gru_seq_bwd(_p(dout), _p(pre), W.data_ptr() + 4 * I, K, _p(gamma), _p(beta), _p(h0),
                                          _p(mask) if ctx.has_mask else None, _p(out), _p(hm) if ctx.has_mask else None, _p(mean), _p(rstd),
                                          _p(dpre), _p(dha), _p(dhb), _p(pa), _p(pb), S, _p(gb[0]), _p(gb[1]), int(direct), _p(ws), _p(ws2), nws,
                                          T, B, D, ctypes.addressof(fin), ctypes.addressof(fin) + 4, _stream()), 'gru_seq_bwd')
            nxt = dha if fin[0] == 0 else dhb
            pnxt = (pa if fin[1] == 0 else pb) if S else None
        for t in (() if seq_c else range(T - 1, -1, -1)):
            if ctx.has_mask:
                hprev, hoff = hm, t * BD
            else:
                hprev, hoff = (h0, 0) if t == 0 else (out, (t - 1) * BD)
            # upstream = dout[t] + d(hm_{t+1}) * mask[t+1]
            _gru_bwd_raw(dout.data_ptr() + 4 * t * BD, nxt.data_ptr() if nxt is not None else None,
                         (mask.data_ptr() + 4 * (t + 1) * B) if (nxt is not None and ctx.has_mask) else None,
                         pre.data_ptr() + 4 * t * B3D, hprev.data_ptr() + 4 * hoff, gamma, beta,
                         mean.data_ptr() + 4 * t * B, rstd.data_ptr() + 4 * t * B, dpre.data_ptr() + 4 * t * B3D,
                         cur.data_ptr(), gb[0], gb[1], ws, B, D,
                         (0 if t == T - 1 else 2) | (4 if t > 0 else 0) | (1 if (direct and t == 0) else 0),
                         pnxt.data_ptr() if (S and pnxt is not None) else None, S if pnxt is not None else 0, BD)
            if S:
                check(lib().genrl_sgemm_skinny_parts(dpre.data_ptr() + 4 * t * B3D, 3 * D, W.data_ptr() + 4 * I, 1, K,
                                                     pcur.data_ptr(), D, BD, B, D, 3 * D, S, _stream()), 'sgemm_parts')
            else:
                sgemm(dpre, 3 * D, 1, W, 1, K, cur, D, None, B, D, 3 * D, accumulate=True, a_off=t * B3D, b_off=I)
            nxt, cur = cur, (dhb if cur is dha else dha)
            if S:
                pnxt, pcur = pcur, (pb if pcur is pa else pa)
        if S:        # d(hm_0) = direct part + slabs
            nxt = nxt + pnxt.sum(0)
        dx = dW = dh0 = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            sgemm(dpre, 3 * D, 1, W, 1, K, dx, I, None, T * B, I, 3 * D)
        if ctx.needs_input_grad[3]:
            tw = _grad_buf(W)
            acc = tw is not None
            dW = tw if acc else torch.empty(3 * D, K, device=dev)
            sgemm(dpre, 1, 3 * D, x, 1, I, dW, K, None, 3 * D, I, T * B, accumulate=acc)
            if ctx.has_mask:
                sgemm(dpre, 1, 3 * D, hm, 1, D, dW, K, None, 3 * D, D, T * B, c_off=I, accumulate=acc)
            else:   # h_{t-1} = [h0, out[:-1]]
                sgemm(dpre, 1, 3 * D, h0, 1, D, dW, K, None, 3 * D, D, B, c_off=I, accumulate=acc)
                if T > 1:
                    ws2 = dW.new_empty(0)
                    nws = lib().genrl_sgemm_ws_floats(3 * D, D, (T - 1) * B)
                    ws2 = torch.empty(nws, device=dev) if nws > 0 else None
                    check(lib().genrl_sgemm(dpre.data_ptr() + 4 * B3D, 1, 3 * D, out.data_ptr(), 1, D,
                                            dW.data_ptr() + 4 * I, K, None, 3 * D, D, (T - 1) * B, 1, _p(ws2), nws,
                                            _stream()), 'sgemm')
            if acc:
                dW = None
        if ctx.needs_input_grad[2]:
            dh0 = nxt * mask[0].unsqueeze(-1) if ctx.has_mask else nxt.clone()
        return (dx, None, dh0, dW, None, None) if direct else (dx, None, dh0, dW, gb[0], gb[1])


def gru_seq(x, mask, h0, W, gamma, beta):
    return _GRUSeq.apply(x, mask, h0, W, gamma, beta)



class _DenseLNAct(Function):
    """y = SiLU(LayerNorm([x1, x2] W^T + b)) as one autograd node (Linear + NormLayer + act,
    agent/dreamer_utils.py:739-747,462-463).  The backward runs the LayerNorm backward once and gets
    dgamma, dbeta AND the Linear's bias gradient from the same pass, then the dgrad / wgrad GEMMs."""
    @staticmethod
    def forward(ctx, x1, x2, W, b, gamma, beta, eps):
        a = _f32(x1).reshape(-1, x1.shape[-1]).contiguous()
        c = _f32(x2).reshape(-1, x2.shape[-1]).contiguous() if x2 is not None else None
        M, K1 = a.shape
        K2 = c.shape[1] if c is not None else 0
        N, K = W.shape
        assert K == K1 + K2
        pre = torch.empty(M, N, device=a.device)
        w1, ld1 = _aligned_block(W, K1, M)
        sgemm(a, K1, 1, w1, ld1, 1, pre, N, b, M, N, K1)
        if c is not None:
            sgemm(c, K2, 1, W, K, 1, pre, N, None, M, N, K2, accumulate=True, b_off=K1)
        ctx.w1 = (w1, ld1)
        y = torch.empty_like(pre)
        mean = torch.empty(M, device=a.device); rstd = torch.empty(M, device=a.device)
        check(lib().genrl_ln_act_fwd(_p(pre), N, _p(gamma), _p(beta), _p(y), N, _p(mean), _p(rstd), M, N, eps, 1,
                                     _stream()), 'ln_act_fwd')
        ctx.save_for_backward(a, c if c is not None else a.new_empty(0), W, gamma, beta, pre, mean, rstd)
        ctx.has2 = c is not None
        ctx.has_bias = b is not None
        ctx.bias = b
        ctx.shapes = (x1.shape, x2.shape if x2 is not None else None)
        return y.reshape(*x1.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        a, c, W, gamma, beta, pre, mean, rstd = ctx.saved_tensors
        M, K1 = a.shape
        K2 = c.shape[1] if ctx.has2 else 0
        N, K = W.shape
        dev = dy.device
        dy2 = dy.reshape(M, N).contiguous()
        dpre = torch.empty_like(pre)          # gradient w.r.t. the pre-LayerNorm projection
        need_p = ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        tg, tb, tc = _grad_buf(gamma), _grad_buf(beta), (_grad_buf(ctx.bias) if ctx.has_bias else None)
        direct = need_p and tg is not None and tb is not None and tc is not None
        if direct:
            g0, g1, g2, acc_p = tg, tb, tc, 1
        elif need_p:
            gb = torch.empty(3, N, device=dev)
            g0, g1, g2, acc_p = gb[0], gb[1], gb[2], 0
        else:
            g0 = g1 = g2 = None; acc_p = 0
        ws = _ws(lib().genrl_ln_ws_floats(M, N), dev) if need_p else None
        if direct:
            acc_p |= defer_reduce(M, N, ws, g0, g1, g2)
        check(lib().genrl_ln_act_bwd(_p(dy2), N, _p(pre), N, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dpre), N,
                                     _p(g0), _p(g1), _p(g2), _p(ws), M, N, 1, acc_p, _stream()), 'ln_act_bwd')
        d1 = d2 = dW = None
        if ctx.needs_input_grad[0]:
            d1 = torch.empty(M, K1, device=dev)
            sgemm(dpre, N, 1, ctx.w1[0], 1, ctx.w1[1], d1, K1, None, M, K1, N)
            d1 = d1.reshape(ctx.shapes[0])
        if ctx.has2 and ctx.needs_input_grad[1]:
            d2 = torch.empty(M, K2, device=dev)
            sgemm(dpre, N, 1, W, 1, K, d2, K2, None, M, K2, N, b_off=K1)
            d2 = d2.reshape(ctx.shapes[1])
        if ctx.needs_input_grad[2]:
            tgt = _grad_buf(W)
            acc = tgt is not None
            if not acc:
                dW = tgt = torch.empty(N, K, device=dev)
            def wg():
                sgemm(dpre, 1, N, a, 1, K1, tgt, K, None, N, K1, M, accumulate=acc)
                if ctx.has2:
                    sgemm(dpre, 1, N, c, 1, K2, tgt, K, None, N, K2, M, accumulate=acc, c_off=K1)
            if acc:
                wgrad_stream.run(wg, dpre, a, c)
            else:
                wg()
        if need_p and not direct:
            return d1, d2, dW, (g2 if ctx.has_bias else None), g0, g1, None
        return d1, d2, dW, None, None, None, None


def dense_ln_act(x1, x2, W, b, gamma, beta, eps=1e-5):
    return _DenseLNAct.apply(x1, x2, W, b, gamma, beta, float(eps))


# ------------------------------------------------------------------ RSSM observe scan with the posterior inside the recurrence

class _ObserveArgs(ctypes.Structure):          # genrl_observe (include/genrl_hip.h)
    _fields_ = ([(n, _CI) for n in ('T', 'B', 'S', 'K', 'D', 'U')] + [(n, _CF) for n in ('unimix', 'in_eps', 'out_eps')]
                + [('w_in_s', _FP), ('ld_in_s', ctypes.c_long), ('w_g', _FP), ('ld_g', ctypes.c_long), ('w_o', _FP),
                   ('ld_o', ctypes.c_long), ('w_d', _FP)]
                + [(n, _FP) for n in ('in_g', 'in_be', 'gru_g', 'gru_be', 'out_g', 'out_be', 'dist_b', 'mask', 'q')]
                + [('out_b', _FP), ('opre_acc', _CI), ('idx', _FP), ('w_in_sT', _FP), ('fuse_sample', _CI)]
                + [(n, _FP) for n in ('sm', 'xpre', 'xh', 'gpre', 'deter', 'opre', 'o', 'plog', 'pst',
                                      'xm', 'xr', 'gm', 'gr', 'om', 'orr', 'ws')]
                + [('ws_floats', ctypes.c_long)]
                + [(n, _FP) for n in ('d_pst', 'dlg', 'dd', 'dov', 'dopre', 'dgpre', 'dxh', 'dxpre', 'dsa', 'dsb', 'dhd_a', 'dhd_b',
                                      'dgamma', 'dbeta', 'gws')]
                + [('direct', _CI)])


def _sgemm_ptr(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, acc, ws, nws):
    """genrl_sgemm on raw addresses (the Python twins of the C launch loops; timed like sgemm() under bench.py's event pass)"""
    if gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    check(lib().genrl_sgemm(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, int(acc), ws, nws, _stream()), 'sgemm')
    if gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        gemm_profile.append((M, N, K, e0, e1, ('k' if a_ks == 1 else 'r') + ('k' if b_ks == 1 else 'r') +
                             ('/skinny' if (M <= 32 and a_ks == 1) else f'/pipe{lib().genrl_sgemm_last_pipe()}')))


def _observe_fwd_py(a):
    """csrc/seq.hip::genrl_observe_seq_fwd, launch for launch (a: _ObserveArgs)"""
    L, st = lib(), _stream()
    T, B, D, U, S, K = a.T, a.B, a.D, a.U, a.S, a.K
    SK, X, f = S * K, U + D, 4
    for t in range(T):
        b0 = t * B
        nxt = t + 1 < T
        if a.idx and a.w_in_sT and t > 0:
            check(L.genrl_onehot_gather_ln_fwd(a.idx + f * b0 * S, S, K, a.w_in_sT, U, a.xpre + f * b0 * U, U, a.in_g, a.in_be,
                                               a.xh + f * b0 * X, X, a.xm + f * b0, a.xr + f * b0, B, U, a.in_eps, st), 'onehot_gather_ln')
        else:
            _sgemm_ptr(a.sm + f * b0 * SK, SK, 1, a.w_in_s, a.ld_in_s, 1, a.xpre + f * b0 * U, U, None, B, U, SK, 1, a.ws, a.ws_floats)
            check(L.genrl_ln_act_fwd(a.xpre + f * b0 * U, U, a.in_g, a.in_be, a.xh + f * b0 * X, X, a.xm + f * b0, a.xr + f * b0, B, U,
                                     a.in_eps, 1, st), 'ln_act_fwd')
        _sgemm_ptr(a.xh + f * b0 * X, X, 1, a.w_g, a.ld_g, 1, a.gpre + f * b0 * 3 * D, 3 * D, None, B, 3 * D, X, 0, a.ws, a.ws_floats)
        check(L.genrl_gru_gates_fwd_ld2(a.gpre + f * b0 * 3 * D, a.xh + f * (b0 * X + U), X, a.gru_g, a.gru_be, a.deter + f * b0 * D, D,
                                        (a.xh + f * ((b0 + B) * X + U)) if nxt else None, X,
                                        (a.mask + f * (b0 + B)) if (nxt and a.mask) else None,
                                        a.gm + f * b0, a.gr + f * b0, B, D, 1e-5, st), 'gru_gates_fwd')
        _sgemm_ptr(a.deter + f * b0 * D, D, 1, a.w_o, a.ld_o, 1, a.opre + f * b0 * U, U, None if a.opre_acc else a.out_b, B, U, D,
                   1 if a.opre_acc else 0, a.ws, a.ws_floats)
        check(L.genrl_ln_act_fwd(a.opre + f * b0 * U, U, a.out_g, a.out_be, a.o + f * b0 * U, U, a.om + f * b0, a.orr + f * b0, B, U,
                                 a.out_eps, 1, st), 'ln_act_fwd')
        idx_next = (a.idx + f * (b0 + B) * S) if (nxt and a.idx) else None
        m_next = (a.mask + f * (b0 + B)) if (nxt and a.mask) else None
        if a.fuse_sample and K == 32:
            check(L.genrl_linear_sample32(a.o + f * b0 * U, U, a.w_d, U, a.dist_b, a.plog + f * b0 * SK, SK, (a.q + f * b0 * SK) if a.q else None,
                                          a.pst + f * b0 * SK, (a.sm + f * (b0 + B) * SK) if nxt else None, idx_next, m_next, B, S, U,
                                          a.unimix, st), 'linear_sample32')
        else:
            _sgemm_ptr(a.o + f * b0 * U, U, 1, a.w_d, U, 1, a.plog + f * b0 * SK, SK, a.dist_b, B, SK, U, 0, a.ws, a.ws_floats)
            check(L.genrl_onehot_fwd_masked(a.plog + f * b0 * SK, (a.q + f * b0 * SK) if a.q else None, a.pst + f * b0 * SK,
                                            (a.sm + f * (b0 + B) * SK) if nxt else None, idx_next, m_next, S, B * S, K, a.unimix, st),
                  'onehot_fwd')


def _observe_bwd_py(a):
    """csrc/seq.hip::genrl_observe_seq_bwd, launch for launch; -> index (0 / 1) of the ping-pong pair that holds step 0's results"""
    L, st = lib(), _stream()
    T, B, D, U, S, K = a.T, a.B, a.D, a.U, a.S, a.K
    SK, X, f = S * K, U + D, 4
    dsm_cur, dsm_nxt, dhd_cur, dhd_nxt = a.dsa, None, a.dhd_a, None
    for t in range(T - 1, -1, -1):
        b0 = t * B
        m1 = a.mask + f * (b0 + B)
        check(L.genrl_onehot_bwd_masked(a.plog + f * b0 * SK, (a.d_pst + f * b0 * SK) if a.d_pst else None, dsm_nxt, m1 if dsm_nxt else None,
                                        S, a.dlg + f * b0 * SK, B * S, K, a.unimix, 1, st), 'onehot_bwd')
        _sgemm_ptr(a.dlg + f * b0 * SK, SK, 1, a.w_d, 1, U, a.dov + f * b0 * U, U, None, B, U, SK, 0, a.ws, a.ws_floats)
        check(L.genrl_ln_act_bwd(a.dov + f * b0 * U, U, a.opre + f * b0 * U, U, a.out_g, a.out_be, a.om + f * b0, a.orr + f * b0,
                                 a.dopre + f * b0 * U, U, None, None, None, None, B, U, 1, 0, st), 'ln_act_bwd')
        _sgemm_ptr(a.dopre + f * b0 * U, U, 1, a.w_o, 1, a.ld_o, a.dd + f * b0 * D, D, None, B, D, U, 1, a.ws, a.ws_floats)
        acc = (0 if t == T - 1 else 2) | (4 if t > 0 else 0) | (1 if (a.direct and t == 0) else 0)
        check(L.genrl_gru_gates_bwd_ldp(a.dd + f * b0 * D, D, dhd_nxt, m1 if dhd_nxt else None, a.gpre + f * b0 * 3 * D,
                                        a.xh + f * (b0 * X + U), X, a.gru_g, a.gru_be, a.gm + f * b0, a.gr + f * b0, a.dgpre + f * b0 * 3 * D,
                                        dhd_cur, D, a.dgamma, a.dbeta, a.gws, B, D, acc,
                                        (a.dxh + f * ((b0 + B) * X + U)) if dhd_nxt else None, 1 if dhd_nxt else 0, 0, X, st), 'gru_gates_bwd')
        _sgemm_ptr(a.dgpre + f * b0 * 3 * D, 3 * D, 1, a.w_g, 1, a.ld_g, a.dxh + f * b0 * X, X, None, B, X, 3 * D, 0, a.ws, a.ws_floats)
        check(L.genrl_ln_act_bwd(a.dxh + f * b0 * X, X, a.xpre + f * b0 * U, U, a.in_g, a.in_be, a.xm + f * b0, a.xr + f * b0,
                                 a.dxpre + f * b0 * U, U, None, None, None, None, B, U, 1, 0, st), 'ln_act_bwd')
        _sgemm_ptr(a.dxpre + f * b0 * U, U, 1, a.w_in_s, 1, a.ld_in_s, dsm_cur, SK, None, B, SK, U, 0, a.ws, a.ws_floats)
        dsm_nxt, dsm_cur = dsm_cur, (a.dsb if dsm_cur == a.dsa else a.dsa)
        dhd_nxt, dhd_cur = dhd_cur, (a.dhd_b if dhd_cur == a.dhd_a else a.dhd_a)
    return 0 if dsm_nxt == a.dsa else 1


def _ln_params_batched(dy, lddy, pre, gamma, beta, bias, mean, rstd, M, N):
    """dgamma / dbeta / bias gradient of a LayerNorm(+SiLU) layer over all M rows at once (the scan's per-step backward launches ask
    for dx only); added straight into the flat gradient buffers under the Optimizer, else returned as (dgamma, dbeta, dbias)"""
    dev = pre.device
    tg, tb, tc = _grad_buf(gamma), _grad_buf(beta), _grad_buf(bias)
    direct = tg is not None and tb is not None and tc is not None
    if direct:
        g0, g1, g2, acc_p = tg, tb, tc, 1
    else:
        gb = torch.empty(3, N, device=dev)
        g0, g1, g2, acc_p = gb[0], gb[1], gb[2], 0
    ws = _ws(lib().genrl_ln_ws_floats(M, N), dev)
    if direct:
        acc_p |= defer_reduce(M, N, ws, g0, g1, g2)
    scratch = torch.empty(M, N, device=dev)
    check(lib().genrl_ln_act_bwd(dy.data_ptr(), lddy, _p(pre), N, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(scratch), N,
                                 _p(g0), _p(g1), _p(g2), _p(ws), M, N, 1, acc_p, _stream()), 'ln_act_bwd')
    return (None, None, None) if direct else (g0, g1, g2)


class _ObserveSeq(Function):
    """EnsembleRSSM.observe WITHOUT single_obs_posterior (conf/defaults/dreamer_v3.yaml:5; agent/dreamer_utils.py:362-371 static_scan over
    obs_step :432-441): the posterior reads [deter_t, embed_t], so the latent sample is part of the recurrence.  ONE autograd node:
    what does not feed the recurrence is batched over T (the action half of _img_in, the embed half of _obs_out; the prior head runs on the
    returned deter outside); the remaining chain is eight dependent launches per step each way, run from C (csrc/seq.hip:
    genrl_observe_seq_fwd / _bwd).  emb (T,B,E), act (T,B,A), mask (T,B) = 1 - is_first, stoch0 (B,S*K), deter0 (B,D), q (T,B*S,K) Exp(1) noise.
    -> deter (T,B,D), post logits (T,B,S*K), post sample (T,B,S*K)."""
    @staticmethod
    def forward(ctx, emb, act, mask, stoch0, deter0, q, W_in, b_in, g_in, be_in, W_g, g_g, be_g, W_o, b_o, g_o, be_o, W_d, b_d,
                eps_in, eps_o):
        emb = _f32(emb).contiguous(); act = _f32(act).contiguous(); mask = _f32(mask).contiguous()
        stoch0 = _f32(stoch0).contiguous(); deter0 = _f32(deter0).contiguous(); q = _f32(q).contiguous()
        T, B, E = emb.shape
        A = act.shape[2]
        D = deter0.shape[1]
        U, Kin = W_in.shape
        SK = Kin - A
        K = q.shape[-1]
        S = SK // K
        X = U + D
        dev = emb.device
        assert W_g.shape == (3 * D, X) and W_o.shape == (U, D + E) and W_d.shape == (SK, U) and stoch0.shape == (B, SK), 'observe_seq shapes'
        assert q.numel() == T * B * SK and U % 4 == 0 and D % 4 == 0 and SK % 4 == 0
        # batched over T: action half of _img_in (+ bias), embed half of _obs_out (+ bias)
        am = torch.empty(T * B, A, device=dev)
        copy2d(act, A, am, A, T * B, A, mask.reshape(T * B))
        xpre = torch.empty(T, B, U, device=dev)
        sgemm(am, A, 1, W_in, Kin, 1, xpre, U, b_in, T * B, U, A, b_off=SK)
        opre = torch.empty(T, B, U, device=dev)
        sgemm(emb, E, 1, W_o, D + E, 1, opre, U, b_o, T * B, U, E, b_off=D)
        # the latent block of _img_in, rows 16-byte aligned (SK + A is not a multiple of 4 with 6 or 10 actions)
        if Kin % 4:
            w_s = torch.empty(U, SK, device=dev)
            copy2d(W_in, Kin, w_s, SK, U, SK)
            ld_s = SK
        else:
            w_s, ld_s = W_in, Kin
        sm = torch.empty(T, B, SK, device=dev)
        copy2d(stoch0, SK, sm, SK, B, SK, mask[0])
        xh = torch.empty(T, B, X, device=dev)
        copy2d(deter0, D, xh, X, B, D, mask[0], dst_off=U)
        gpre = torch.empty(T, B, 3 * D, device=dev)
        deter = torch.empty(T, B, D, device=dev)
        o = torch.empty(T, B, U, device=dev)
        plog = torch.empty(T, B, SK, device=dev)
        pst = torch.empty(T, B, SK, device=dev)
        stats = torch.empty(6, T, B, device=dev)
        a = _ObserveArgs()
        a.T, a.B, a.S, a.K, a.D, a.U = T, B, S, K, D, U
        a.unimix, a.in_eps, a.out_eps = UNIMIX, eps_in, eps_o
        a.w_in_s, a.ld_in_s, a.w_g, a.ld_g, a.w_o, a.ld_o, a.w_d = _p(w_s), ld_s, _p(W_g), X, _p(W_o), D + E, _p(W_d)
        a.opre_acc = 1
        for n_, t_ in (('in_g', g_in), ('in_be', be_in), ('gru_g', g_g), ('gru_be', be_g), ('out_g', g_o), ('out_be', be_o), ('dist_b', b_d),
                       ('mask', mask), ('q', q), ('sm', sm), ('xpre', xpre), ('xh', xh), ('gpre', gpre), ('deter', deter), ('opre', opre),
                       ('o', o), ('plog', plog), ('pst', pst)):
            setattr(a, n_, _p(t_))
        for i, n_ in enumerate(('xm', 'xr', 'gm', 'gr', 'om', 'orr')):
            setattr(a, n_, stats[i].data_ptr())
        nws = max(lib().genrl_sgemm_ws_floats(*s_) for s_ in _observe_shapes(B, SK, U, D))
        ws = torch.empty(max(nws, 1), device=dev)
        a.ws, a.ws_floats = ws.data_ptr(), nws
        keep = _scan_fuse(a, w_s, T, B, S, K, U, dev)
        if SEQ_C and gemm_profile is None:
            check(lib().genrl_observe_seq_fwd(ctypes.byref(a), _stream()), 'observe_seq_fwd')
        else:
            _observe_fwd_py(a)
        del keep
        ctx.save_for_backward(emb, am, mask, q, W_in, w_s, b_in, g_in, be_in, W_g, g_g, be_g, W_o, b_o, g_o, be_o, W_d, b_d,
                              sm, xpre, xh, gpre, deter, opre, o, plog, stats)
        ctx.dims = (T, B, S, K, D, U, A, E)
        ctx.eps = (eps_in, eps_o)
        return deter, plog, pst

    @staticmethod
    def backward(ctx, d_deter, d_plog, d_pst):
        (emb, am, mask, q, W_in, w_s, b_in, g_in, be_in, W_g, g_g, be_g, W_o, b_o, g_o, be_o, W_d, b_d,
         sm, xpre, xh, gpre, deter, opre, o, plog, stats) = ctx.saved_tensors
        T, B, S, K, D, U, A, E = ctx.dims
        SK, X, Kin, M = S * K, U + D, S * K + A, T * B
        dev = emb.device
        dd = d_deter.contiguous().clone() if d_deter is not None else torch.zeros(T, B, D, device=dev)
        dlg = d_plog.contiguous().clone() if d_plog is not None else torch.zeros(T, B, SK, device=dev)
        d_pst = d_pst.contiguous() if d_pst is not None else None
        dov = torch.empty(T, B, U, device=dev); dopre = torch.empty(T, B, U, device=dev)
        dgpre = torch.empty(T, B, 3 * D, device=dev); dxh = torch.empty(T, B, X, device=dev); dxpre = torch.empty(T, B, U, device=dev)
        ds2 = torch.empty(2, B, SK, device=dev); dh2 = torch.empty(2, B, D, device=dev)
        tg, tb = _grad_buf(g_g), _grad_buf(be_g)
        direct = tg is not None and tb is not None
        gb = (tg, tb) if direct else torch.empty(2, 3 * D, device=dev)
        gws = torch.empty(lib().genrl_gru_ws_floats(B, D), device=dev)
        a = _ObserveArgs()
        a.T, a.B, a.S, a.K, a.D, a.U = T, B, S, K, D, U
        a.unimix, a.in_eps, a.out_eps = UNIMIX, ctx.eps[0], ctx.eps[1]
        a.w_in_s, a.ld_in_s, a.w_g, a.ld_g, a.w_o, a.ld_o, a.w_d = _p(w_s), w_s.shape[1], _p(W_g), X, _p(W_o), D + E, _p(W_d)
        a.opre_acc = 1
        for n_, t_ in (('in_g', g_in), ('in_be', be_in), ('gru_g', g_g), ('gru_be', be_g), ('out_g', g_o), ('out_be', be_o), ('dist_b', b_d),
                       ('mask', mask), ('q', q), ('sm', sm), ('xpre', xpre), ('xh', xh), ('gpre', gpre), ('deter', deter), ('opre', opre),
                       ('o', o), ('plog', plog), ('d_pst', d_pst), ('dlg', dlg), ('dd', dd), ('dov', dov), ('dopre', dopre),
                       ('dgpre', dgpre), ('dxh', dxh), ('dxpre', dxpre), ('dgamma', gb[0]), ('dbeta', gb[1]), ('gws', gws)):
            setattr(a, n_, _p(t_))
        for i, n_ in enumerate(('xm', 'xr', 'gm', 'gr', 'om', 'orr')):
            setattr(a, n_, stats[i].data_ptr())
        a.dsa, a.dsb, a.dhd_a, a.dhd_b = ds2[0].data_ptr(), ds2[1].data_ptr(), dh2[0].data_ptr(), dh2[1].data_ptr()
        a.direct = int(direct)
        nws = max(lib().genrl_sgemm_ws_floats(*s_) for s_ in _observe_shapes(B, SK, U, D))
        ws = torch.empty(max(nws, 1), device=dev)
        a.ws, a.ws_floats = ws.data_ptr(), nws
        if SEQ_C and gemm_profile is None:
            fin = ctypes.c_int(0)
            check(lib().genrl_observe_seq_bwd(ctypes.byref(a), ctypes.byref(fin), _stream()), 'observe_seq_bwd')
            fin = fin.value
        else:
            fin = _observe_bwd_py(a)
        need = ctx.needs_input_grad
        # ---- batched over T: weight gradients, LayerNorm parameters, the encoder's gradient
        def wgrad(P, parts):
            """dP (+)= sum over parts of dY^T X into column blocks: parts = [(dY2d, ldy, N, X2d, ldx, Kx, c_off)]"""
            if P is None:
                return None
            tgt = _grad_buf(P)
            acc = tgt is not None
            out = tgt if acc else torch.empty_like(P)
            for (dy_, ldy_, n_, x_, ldx_, k_, off_) in parts:
                sgemm(dy_, 1, ldy_, x_, 1, ldx_, out, P.shape[1], None, n_, k_, M, accumulate=acc, c_off=off_)
            return None if acc else out
        dW_d = wgrad(W_d if need[17] else None, [(dlg, SK, SK, o, U, U, 0)])
        db_d = None
        if need[18]:
            tgt = _grad_buf(b_d)
            if tgt is not None:
                colsum(dlg.reshape(M, SK), out=tgt, accumulate=True)
            else:
                db_d = colsum(dlg.reshape(M, SK))
        dg_o, dbe_o, db_o = _ln_params_batched(dov, U, opre, g_o, be_o, b_o, stats[4], stats[5], M, U)
        dW_o = wgrad(W_o if need[13] else None, [(dopre, U, U, deter, D, D, 0), (dopre, U, U, emb, E, E, D)])
        d_emb = None
        if need[0]:
            d_emb = torch.empty(T, B, E, device=dev)
            sgemm(dopre, U, 1, W_o, 1, D + E, d_emb, E, None, M, E, U, b_off=D)
        dW_g = wgrad(W_g if need[10] else None, [(dgpre, 3 * D, 3 * D, xh, X, X, 0)])
        dg_i, dbe_i, db_i = _ln_params_batched(dxh, X, xpre, g_in, be_in, b_in, stats[0], stats[1], M, U)
        dW_in = wgrad(W_in if need[6] else None, [(dxpre, U, U, sm, SK, SK, 0), (dxpre, U, U, am, A, A, SK)])
        d_act = None
        if need[1]:
            d_act = torch.empty(T, B, A, device=dev)
            sgemm(dxpre, U, 1, W_in, 1, Kin, d_act, A, None, M, A, U, b_off=SK)
            d_act = d_act * mask.unsqueeze(-1)
        d_s0 = (ds2[fin] * mask[0].unsqueeze(-1)) if need[3] else None
        d_h0 = ((dh2[fin] + dxh[0, :, U:]) * mask[0].unsqueeze(-1)) if need[4] else None
        dgg, dbg = (None, None) if direct else (gb[0], gb[1])
        return (d_emb, d_act, None, d_s0, d_h0, None, dW_in, db_i, dg_i, dbe_i, dW_g, dgg, dbg, dW_o, db_o, dg_o, dbe_o, dW_d, db_d,
                None, None)


OBSERVE_FUSE = os.environ.get('GENRL_OBSERVE_FUSE', '1') != '0'     # the scans' fused forward launches (gather + LayerNorm, head product + sample)


def _scan_fuse(a, w_s, T, B, S, K, U, dev):
    """fill the fused-forward fields of an _ObserveArgs (csrc/seq.hip: six instead of eight launches per step); -> tensors to keep alive"""
    if not OBSERVE_FUSE or gemm_precision_is_p16():
        return ()
    keep = []
    if U % 4 == 0 and U <= 1024 and S <= 64 and T > 1:
        if w_s.shape[1] != S * K:               # (the caller's latent block is a column range of the wider weight: compact it first)
            w_c = torch.empty(U, S * K, device=dev)
            copy2d(w_s, w_s.shape[1], w_c, S * K, U, S * K)
            w_s = w_c
        w_sT = transpose_last2_raw(w_s.reshape(1, U, S * K)).reshape(S * K, U)        # [S K][U]: a latent class = one row
        idx = torch.empty(T, B, S, dtype=torch.int32, device=dev)
        a.idx, a.w_in_sT = idx.data_ptr(), w_sT.data_ptr()
        keep += [w_sT, idx]
    if K == 32 and U % 16 == 0:
        a.fuse_sample = 1
    return tuple(keep)


def gemm_precision_is_p16():
    return lib().genrl_gemm_precision() == 1


def _observe_shapes(B, SK, U, D):
    X = U + D
    return [(B, U, SK), (B, 3 * D, X), (B, U, D), (B, SK, U), (B, D, U), (B, X, 3 * D)]


def observe_seq(emb, act, mask, stoch0, deter0, q, W_in, b_in, g_in, be_in, W_g, g_g, be_g, W_o, b_o, g_o, be_o, W_d, b_d,
                eps_in=1e-5, eps_o=1e-5):
    return _ObserveSeq.apply(emb, act, mask, stoch0, deter0, q, W_in, b_in, g_in, be_in, W_g, g_g, be_g, W_o, b_o, g_o, be_o, W_d, b_d,
                             float(eps_in), float(eps_o))


def rssm_imagine_seq(act, stoch0, deter0, q, S, K, W_in, b_in, g_in, be_in, W_g, g_g, be_g, W_out, b_out, g_out, be_out, W_dist,
                     b_dist, eps_in=1e-5, eps_out=1e-5):
    """EnsembleRSSM.imagine for GIVEN actions (agent/dreamer_utils.py:373-381: static_scan over img_step :459-473), forward only (no
    autograd graph: the data-free block's warm-up rollouts and the report / video_imagine paths run under no_grad).  The action half
    of _img_in is batched over T; the remaining chain -- eight launches per step -- runs from ONE host call (genrl_observe_seq_fwd
    with the prior head in the posterior's place).  act (T,B,A) time-major, stoch0 (B,S*K), deter0 (B,D), q (T,B*S,K) Exp(1) noise or
    None for mode().  -> deter (T,B,D), logit (T,B,S*K), stoch (T,B,S*K)."""
    act = _f32(act).contiguous(); stoch0 = _f32(stoch0).contiguous(); deter0 = _f32(deter0).contiguous()
    q = _f32(q).contiguous() if q is not None else None
    T, B, A = act.shape
    D = deter0.shape[1]
    U, Kin = W_in.shape
    SK, X = S * K, U + D
    dev = act.device
    assert Kin == SK + A and W_g.shape == (3 * D, X) and W_out.shape == (U, D) and W_dist.shape == (SK, U) and stoch0.shape == (B, SK)
    assert U % 4 == 0 and D % 4 == 0 and SK % 4 == 0 and (q is None or q.numel() == T * B * SK)
    xpre = torch.empty(T, B, U, device=dev)
    sgemm(act, A, 1, W_in, Kin, 1, xpre, U, b_in, T * B, U, A, b_off=SK)
    if Kin % 4:                                   # the latent block of _img_in with 16-byte aligned rows
        w_s = torch.empty(U, SK, device=dev)
        copy2d(W_in, Kin, w_s, SK, U, SK)
        ld_s = SK
    else:
        w_s, ld_s = W_in, Kin
    sm = torch.empty(T, B, SK, device=dev)
    copy2d(stoch0, SK, sm, SK, B, SK)
    xh = torch.empty(T, B, X, device=dev)
    copy2d(deter0, D, xh, X, B, D, dst_off=U)
    gpre = torch.empty(T, B, 3 * D, device=dev)
    deter = torch.empty(T, B, D, device=dev)
    opre = torch.empty(T, B, U, device=dev)
    o = torch.empty(T, B, U, device=dev)
    plog = torch.empty(T, B, SK, device=dev)
    pst = torch.empty(T, B, SK, device=dev)
    stats = torch.empty(6, T, B, device=dev)
    a = _ObserveArgs()
    a.T, a.B, a.S, a.K, a.D, a.U = T, B, S, K, D, U
    a.unimix, a.in_eps, a.out_eps = UNIMIX, eps_in, eps_out
    a.w_in_s, a.ld_in_s, a.w_g, a.ld_g, a.w_o, a.ld_o, a.w_d = _p(w_s), ld_s, _p(W_g), X, _p(W_out), D, _p(W_dist)
    a.opre_acc = 0
    for n_, t_ in (('in_g', g_in), ('in_be', be_in), ('gru_g', g_g), ('gru_be', be_g), ('out_g', g_out), ('out_be', be_out),
                   ('dist_b', b_dist), ('out_b', b_out), ('q', q), ('sm', sm), ('xpre', xpre), ('xh', xh), ('gpre', gpre), ('deter', deter),
                   ('opre', opre), ('o', o), ('plog', plog), ('pst', pst)):
        setattr(a, n_, _p(t_))
    for i, n_ in enumerate(('xm', 'xr', 'gm', 'gr', 'om', 'orr')):
        setattr(a, n_, stats[i].data_ptr())
    nws = max(lib().genrl_sgemm_ws_floats(*s_) for s_ in _observe_shapes(B, SK, U, D))
    ws = torch.empty(max(nws, 1), device=dev)
    a.ws, a.ws_floats = ws.data_ptr(), nws
    keep = _scan_fuse(a, w_s, T, B, S, K, U, dev)
    if SEQ_C and gemm_profile is None:
        check(lib().genrl_observe_seq_fwd(ctypes.byref(a), _stream()), 'observe_seq_fwd')
    else:
        _observe_fwd_py(a)
    del keep
    return deter, plog, pst
