"""The imagination rollout (WorldModel.imagine, agent/dreamer.py:254-287) on pre-split ("h2 plane") GEMM operands.

Same autograd node structure as ops._Rollout / ops.ActorTape (one node for the H-step loop, the policy's backward batched
over all H*N rows), but every activation that feeds a GEMM is written by its producing row kernel as two fp16 planes of
the row-scaled value plus the row's inverse scale (genrl_*_h2 entry points) next to its fp32 copy, the frozen world-model /
policy weights are split once per optimiser step (planes.weight), and the products run in genrl_gemm_h2: fp32-accurate
arithmetic on the fp16 matrix cores (three MFMAs per block and k-step) with a pure DMA + MFMA K loop.  Two-input layers ([stoch, action] -> img_in, [x, deter] -> GRU, [stoch, deter] -> policy) are ONE
launch with two operand segments.  Weight gradients stay on the fp32-operand kernels (ops.sgemm)."""
import torch
from torch.autograd import Function

from ._lib import lib, check
from . import planes
from . import planes as pl          # (module alias: `planes=` parameters below are operand handles)
from . import ops
from .ops import _p, _f32, _grad_buf, _ws, sgemm, colsum, UNIMIX

_stream = ops._stream


# ---- genrl_rollout (include/genrl_hip.h): the arguments of the rollout's launch loop in C (csrc/seq.hip: genrl_imagine_seq_fwd)
import ctypes as _ct


class _PRef(_ct.Structure):
    _fields_ = [('p', _ct.c_void_p), ('ld', _ct.c_long), ('plane', _ct.c_long), ('inv', _ct.c_void_p)]


_FP, _F, _I = _ct.c_void_p, _ct.c_float, _ct.c_int


class _RolloutArgs(_ct.Structure):
    _fields_ = ([(n, _I) for n in ('H', 'N', 'S', 'K', 'D', 'A', 'AP', 'U', 'L')] + [(n, _F) for n in ('unimix', 'min_std', 'max_std')]
                + [(n, _FP) for n in ('stoch', 'deter', 'logit', 'action', 'raws', 'eps', 'q')]
                + [(n, _PRef) for n in ('stoch_p', 'deter_p', 'act_p', 'x_p', 'o_p')]
                + [(n, _FP) for n in ('x_pre', 'x', 'g_pre', 'o_pre', 'o', 'xm', 'xr', 'gm', 'gr', 'om', 'orr')]
                + [(n, _PRef) for n in ('w_in_s', 'w_in_a', 'w_g_x', 'w_g_h', 'w_out', 'w_dist')]
                + [('in_b', _FP), ('in_g', _FP), ('in_be', _FP), ('in_eps', _F), ('gru_g', _FP), ('gru_be', _FP),
                   ('out_b', _FP), ('out_g', _FP), ('out_be', _FP), ('out_eps', _F), ('dist_b', _FP),
                   ('pw0s', _PRef), ('pw0d', _PRef), ('pw', _PRef * 8), ('pb', _FP * 8), ('pg', _FP * 8), ('pbe', _FP * 8),
                   ('peps', _F * 8), ('pU', _I * 8), ('ppre', _FP * 8), ('py', _FP * 8), ('pmean', _FP * 8), ('prstd', _FP * 8),
                   ('pyp', _PRef * 8), ('head_w', _FP), ('head_b', _FP), ('ln_part', _FP), ('ln_sync', _FP)])


class _RolloutBwdArgs(_ct.Structure):
    _fields_ = ([(n, _I) for n in ('H', 'N', 'S', 'K', 'D', 'A', 'AP', 'U')] + [(n, _F) for n in ('unimix', 'min_std', 'max_std')]
                + [(n, _FP) for n in ('logit', 'deter', 'raws', 'eps', 'x_pre', 'g_pre', 'o_pre', 'xm', 'xr', 'gm', 'gr', 'om', 'orr',
                                      'ds', 'dd', 'dl_in', 'dact_all', 'd_raw', 'dlg', 'dov', 'do_pre', 'dg_pre', 'dx', 'dx_pre', 'dha', 'dhb')]
                + [(n, _PRef) for n in ('dlg_p', 'dop_p', 'dg_p', 'dxp_p', 'wt_dist', 'wt_out', 'wt_g_x', 'wt_g_h', 'wt_in_s')]
                + [(n, _FP) for n in ('waT', 'out_g', 'out_be', 'gru_g', 'gru_be', 'in_g', 'in_be')])


def _pref(P):
    return _PRef(P.t.data_ptr(), P.ld, P.plane, P.inv.data_ptr())


def _ln_fwd(pre_ptr, gamma, beta, y_ptr, mean_ptr, rstd_ptr, M, N, eps, P, row0):
    check(lib().genrl_ln_act_fwd_h2(pre_ptr, N, _p(gamma), _p(beta), y_ptr, N, mean_ptr, rstd_ptr, M, N, eps, 1,
                                    P.ptr(row0), P.ld, P.plane, P.inv_ptr(row0), _stream()), 'ln_act_fwd_h2')


def _ln_bwd(dy_ptr, pre_ptr, gamma, beta, mean_ptr, rstd_ptr, dpre_ptr, M, N, P, row0, g0=None, g1=None, g2=None, ws=None,
            acc_p=0):
    check(lib().genrl_ln_act_bwd_h2(dy_ptr, N, pre_ptr, N, _p(gamma), _p(beta), mean_ptr, rstd_ptr, dpre_ptr, N, _p(g0),
                                    _p(g1), _p(g2), _p(ws), M, N, 1, acc_p, P.ptr(row0), P.ld, P.plane, P.inv_ptr(row0),
                                    _stream()), 'ln_act_bwd_h2')


class ActorTapePlanes(ops.ActorTape):
    """ops.ActorTape with plane copies of the hidden activations (time-major rows h*N + n) and plane-operand products for the
    forward and the batched dgrad; the weight gradients read the fp32 copies."""
    def __init__(self, H, N, layers, head_w, head_b, dev):
        super().__init__(H, N, layers, head_w, head_b, dev)
        self.yp = [planes.Planes(H * N, l[0].shape[0], dev) for l in layers]

    def _forward_planes(self, t, sp_, dp_, out):
        """layer 0 input = rows t*N.. of the rollout's stoch / deter planes (sp_, dp_)"""
        N = self.N
        Ap, row0 = None, t * N
        for l, (W, b, gamma, beta, eps) in enumerate(self.layers):
            U, K = W.shape
            pre, y = self.pre[l], self.y[l]
            off = t * N * U
            fuse = planes.gemm_ln_ok(N, U, pre.device)
            if l == 0:
                K1 = sp_.cols
                if fuse:
                    planes.gemm_ln(sp_, planes.weight(W, c0=0, c1=K1), pre, b, N, U, gamma, beta, eps, self.yp[l], row0, y=y, mean=self.mean[l],
                                   rstd=self.rstd[l], a_row0=row0, A1=dp_, B1=planes.weight(W, c0=K1), a1_row0=row0, c_off=off, y_off=off, m_off=t * N)
                else:
                    planes.gemm(sp_, planes.weight(W, c0=0, c1=K1), pre, U, b, N, U, a_row0=row0, A1=dp_, B1=planes.weight(W, c0=K1),
                                a1_row0=row0, c_off=off)
            elif fuse:
                planes.gemm_ln(Ap, planes.weight(W), pre, b, N, U, gamma, beta, eps, self.yp[l], row0, y=y, mean=self.mean[l], rstd=self.rstd[l],
                               a_row0=row0, c_off=off, y_off=off, m_off=t * N)
            else:
                planes.gemm(Ap, planes.weight(W), pre, U, b, N, U, a_row0=row0, c_off=off)
            if not fuse:
                _ln_fwd(pre.data_ptr() + 4 * off, gamma, beta, y.data_ptr() + 4 * off, self.mean[l].data_ptr() + 4 * t * N,
                        self.rstd[l].data_ptr() + 4 * t * N, N, U, eps, self.yp[l], row0)
            Ap = self.yp[l]
        if out is None:               # the caller runs the output layer fused with the Normal head (ActorTape.head_fused)
            return None
        A2, Kx = self.head_w.shape
        sgemm(self.y[-1], Kx, 1, self.head_w, Kx, 1, out, A2, self.head_b, N, A2, Kx, a_off=t * N * Kx)
        return out

    def _backward(self):
        assert self.inputs is not None, 'ActorTape.inputs (time-major rollout states) not set'
        H, N = self.H, self.N
        M = H * N
        dev = self.d_raw.device
        A2, U = self.head_w.shape
        d = self.d_raw.reshape(M, A2)
        x_last = self.y[-1]
        dWh, dbh = self.head_param_grads(d, x_last)
        dy = torch.empty(M, U, device=dev)
        sgemm(d, A2, 1, self.head_w, 1, U, dy, U, None, M, U, A2)
        grads = [None] * len(self.layers)
        dpre_p = None
        for l in range(len(self.layers) - 1, -1, -1):
            W, b, gamma, beta, eps = self.layers[l]
            U, K = W.shape
            dpre = torch.empty(M, U, device=dev)
            if dpre_p is None or dpre_p.cols != U:
                dpre_p = planes.Planes(M, U, dev)
            tg, tbe, tc = _grad_buf(gamma), _grad_buf(beta), (_grad_buf(b) if b is not None else None)
            direct = tg is not None and tbe is not None and (b is None or tc is not None)
            if direct:
                g0, g1, g2, acc_p = tg, tbe, tc, 1
            else:
                gb = torch.empty(3, U, device=dev)
                g0, g1, g2, acc_p = gb[0], gb[1], gb[2], 0
            ws = _ws(lib().genrl_ln_ws_floats(M, U), dev)
            if direct:
                acc_p |= ops.defer_reduce(M, U, ws, g0, g1, g2)
            _ln_bwd(_p(dy), _p(self.pre[l]), gamma, beta, _p(self.mean[l]), _p(self.rstd[l]), _p(dpre), M, U, dpre_p, 0,
                    g0, g1, g2, ws, acc_p)
            tw = _grad_buf(W)
            acc = tw is not None
            dW = tw if acc else torch.empty(U, K, device=dev)
            # weight gradients: on the same planes through the transposing kernel (genrl_gemm_h2_tn) from TN_MIN_ROWS rows up
            tn = planes.tn_ok(M, U, K, K)
            if l > 0:
                x = self.y[l - 1]
                if tn:
                    planes.gemm_tn(dpre_p, self.yp[l - 1], dW, K, U, K, M, accumulate=acc)
                else:
                    sgemm(dpre, 1, U, x, 1, K, dW, K, None, U, K, M, accumulate=acc)
                dy = torch.empty(M, K, device=dev)
                planes.gemm(dpre_p, planes.weight(W, transpose=True), dy, K, None, M, K)
            else:
                x1, x2 = self.inputs
                K1, K2 = x1.shape[-1], x2.shape[-1]
                assert x1.is_contiguous() and x2.is_contiguous() and x1.shape[0] >= H and K1 + K2 == K
                sp_ = getattr(self, 'state_planes', None)
                if tn and sp_ is not None and K1 % 4 == 0:
                    planes.gemm_tn(dpre_p, sp_[0], dW, K, U, K1, M, accumulate=acc)
                    planes.gemm_tn(dpre_p, sp_[1], dW, K, U, K2, M, accumulate=acc, c_off=K1)
                else:
                    sgemm(dpre, 1, U, x1, 1, K1, dW, K, None, U, K1, M, accumulate=acc)
                    sgemm(dpre, 1, U, x2, 1, K2, dW, K, None, U, K2, M, c_off=K1, accumulate=acc)
            grads[l] = (None if acc else dW, None if (direct or b is None) else g2, None if direct else g0,
                        None if direct else g1)
        return dWh, dbh, grads


class _RolloutPlanes(Function):
    """ops._Rollout on plane operands.  Returns time-major stoch (H+1,N,S,K), deter (H+1,N,D), logit (H+1,N,S,K),
    action (H+1,N,A), raw (H,N,2A)."""
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
        AP = (A + 3) // 4 * 4
        stoch = f(H + 1, N, SK); deter = f(H + 1, N, D); logit = f(H + 1, N, SK)
        action = torch.zeros(H + 1, N, AP, device=dev)
        raws = f(H, N, 2 * A)
        stoch[0].copy_(stoch0.reshape(N, SK)); deter[0].copy_(deter0); logit[0].copy_(logit0.reshape(N, SK))
        # planes of every GEMM operand, rows h*N + n
        stoch_p, deter_p = planes.Planes((H + 1) * N, SK, dev), planes.Planes((H + 1) * N, D, dev)
        act_p = planes.Planes((H + 1) * N, A, dev)                 # (zero padded to 64 columns; row block 0 unused)
        x_p, o_p = planes.Planes(N, U, dev), planes.Planes(N, U, dev)      # consumed within the step: one row block
        planes.split(stoch[0], out=stoch_p); planes.split(deter[0], out=deter_p)
        # (x and o -- the img_in / img_out activations -- are consumed within the step as PLANES only: no fp32 copy is written where the
        # LayerNorm kernel can do without, 5.9 -> 5.4 us per launch at 1 024 rows)
        planes_only = 256 < U <= 4096 and U % 4 == 0
        x_pre, x = f(H, N, U), (None if planes_only else f(N, U))
        g_pre = f(H, N, 3 * D)
        o_pre, o = f(H, N, U), (None if planes_only else f(N, U))
        st = {k: f(H, N) for k in ('xm', 'xr', 'gm', 'gr', 'om', 'or')}
        eps = _f32(eps).contiguous(); q = _f32(q).contiguous()
        w_in_s, w_in_a = planes.weight(sp.in_w, c0=0, c1=SK), planes.weight(sp.in_w, c0=SK, c1=SK + A)
        w_g_x, w_g_h = planes.weight(sp.gru_w, c0=0, c1=U), planes.weight(sp.gru_w, c0=U)
        w_out, w_dist = planes.weight(sp.out_w), planes.weight(sp.dist_w)
        pt = lambda t, off: t.data_ptr() + 4 * off
        L = lib()
        seq_c = ops.SEQ_C and planes.gemm_profile is None and len(tape.layers) <= 8
        # Dense -> LayerNorm -> SiLU as ONE launch where the shape allows (genrl_gemm_h2_ln: policy layers, img_in, img_out: 10 launches per step)
        fuse_any = planes.ln_fused_here(dev)                 # (per layer: genrl_gemm_h2_ln_ok(N, width), in C and in the Python twin alike)
        fuse_ln = fuse_any and planes.gemm_ln_ok(N, U, dev)
        if seq_c:
            # the H-step launch loop in C (csrc/seq.hip: genrl_imagine_seq_fwd -- the loop below, launch for launch, from one host call)
            a = _RolloutArgs()
            a.H, a.N, a.S, a.K, a.D, a.A, a.AP, a.U, a.L = H, N, S, K, D, A, AP, U, len(tape.layers)
            a.unimix, a.min_std, a.max_std = UNIMIX, sp.min_std, sp.max_std
            for n_, t_ in (('stoch', stoch), ('deter', deter), ('logit', logit), ('action', action), ('raws', raws), ('eps', eps), ('q', q),
                           ('x_pre', x_pre), ('x', x), ('g_pre', g_pre), ('o_pre', o_pre), ('o', o), ('xm', st['xm']), ('xr', st['xr']),
                           ('gm', st['gm']), ('gr', st['gr']), ('om', st['om']), ('orr', st['or']), ('in_b', sp.in_b), ('in_g', sp.in_g),
                           ('in_be', sp.in_be), ('gru_g', sp.gru_g), ('gru_be', sp.gru_be), ('out_b', sp.out_b), ('out_g', sp.out_g),
                           ('out_be', sp.out_be), ('dist_b', sp.dist_b), ('head_w', tape.head_w), ('head_b', tape.head_b)):
                setattr(a, n_, _p(t_))
            a.in_eps, a.out_eps = sp.in_eps, sp.out_eps
            for n_, P_ in (('stoch_p', stoch_p), ('deter_p', deter_p), ('act_p', act_p), ('x_p', x_p), ('o_p', o_p), ('w_in_s', w_in_s),
                           ('w_in_a', w_in_a), ('w_g_x', w_g_x), ('w_g_h', w_g_h), ('w_out', w_out), ('w_dist', w_dist)):
                setattr(a, n_, _pref(P_))
            for l, (W_, b_, ga_, be_, eps_) in enumerate(tape.layers):
                if l == 0:
                    a.pw0s, a.pw0d = _pref(planes.weight(W_, c0=0, c1=SK)), _pref(planes.weight(W_, c0=SK))
                else:
                    a.pw[l] = _pref(planes.weight(W_))
                a.pb[l], a.pg[l], a.pbe[l], a.peps[l], a.pU[l] = _p(b_), _p(ga_), _p(be_), eps_, W_.shape[0]
                a.ppre[l], a.py[l], a.pmean[l], a.prstd[l] = _p(tape.pre[l]), _p(tape.y[l]), _p(tape.mean[l]), _p(tape.rstd[l])
                a.pyp[l] = _pref(tape.yp[l])
            if fuse_any:
                sync_, part_ = planes._ln_workspace(dev)
                a.ln_part, a.ln_sync = (part_.data_ptr() + 15) // 16 * 16, sync_.data_ptr()
            check(L.genrl_imagine_seq_fwd(_ct.addressof(a), _stream()), 'imagine_seq_fwd')
        for h in (() if seq_c else range(H)):
            r0, r1 = h * N, (h + 1) * N
            tape._forward_planes(h, stoch_p, deter_p, None)
            tape.head_fused(h, pt(eps, h * N * A), pt(raws, h * N * 2 * A), pt(action, r1 * AP), AP, sp.min_std, sp.max_std, act_p, r1)
            # img_in: [stoch_h | action_{h+1}] -> hidden, LN + SiLU
            if fuse_ln:
                planes.gemm_ln(stoch_p, w_in_s, x_pre, sp.in_b, N, U, sp.in_g, sp.in_be, sp.in_eps, x_p, 0, y=x, mean=st['xm'], rstd=st['xr'],
                               a_row0=r0, A1=act_p, B1=w_in_a, a1_row0=r1, c_off=h * N * U, m_off=r0)
            else:
                planes.gemm(stoch_p, w_in_s, x_pre, U, sp.in_b, N, U, a_row0=r0, A1=act_p, B1=w_in_a, a1_row0=r1, c_off=h * N * U)
                _ln_fwd(pt(x_pre, h * N * U), sp.in_g, sp.in_be, _p(x), pt(st['xm'], r0), pt(st['xr'], r0), N, U, sp.in_eps, x_p, 0)
            # GRU: [x | deter_h] W_g^T -> LN + gates -> deter_{h+1}
            planes.gemm(x_p, w_g_x, g_pre, 3 * D, None, N, 3 * D, A1=deter_p, B1=w_g_h, a1_row0=r0, c_off=h * N * 3 * D)
            check(L.genrl_gru_gates_fwd_h2(pt(g_pre, h * N * 3 * D), pt(deter, r0 * D), D, _p(sp.gru_g), _p(sp.gru_be),
                                           pt(deter, r1 * D), D, None, None, pt(st['gm'], r0), pt(st['gr'], r0), N, D, 1e-5,
                                           deter_p.ptr(r1), deter_p.ld, deter_p.plane, deter_p.inv_ptr(r1), _stream()),
                  'gru_gates_fwd_h2')
            # prior head: img_out (+LN+SiLU), dist, sample
            if fuse_ln:
                planes.gemm_ln(deter_p, w_out, o_pre, sp.out_b, N, U, sp.out_g, sp.out_be, sp.out_eps, o_p, 0, y=o, mean=st['om'], rstd=st['or'],
                               a_row0=r1, c_off=h * N * U, m_off=r0)
            else:
                planes.gemm(deter_p, w_out, o_pre, U, sp.out_b, N, U, a_row0=r1, c_off=h * N * U)
                _ln_fwd(pt(o_pre, h * N * U), sp.out_g, sp.out_be, _p(o), pt(st['om'], r0), pt(st['or'], r0), N, U, sp.out_eps, o_p, 0)
            if K == 32 and sp.dist_b is not None:
                # prior logits AND their sample in one launch: softmax -> unimix -> exponential race in the product's epilogue
                planes.gemm_sample(o_p, w_dist, logit, SK, sp.dist_b, N, SK, q, SK, UNIMIX, stoch, SK, stoch_p, c_off=r1 * SK,
                                   q_off=h * N * SK, s_off=r1 * SK, sp_row0=r1)
            else:
                planes.gemm(o_p, w_dist, logit, SK, sp.dist_b, N, SK, c_off=r1 * SK)
                check(L.genrl_onehot_fwd_h2(pt(logit, r1 * SK), pt(q, h * N * SK), pt(stoch, r1 * SK), None, N * S, K, UNIMIX,
                                            stoch_p.ptr(r1), SK, stoch_p.ld, stoch_p.plane, stoch_p.inv_ptr(r1), _stream()),
                      'onehot_fwd_h2')
        tape.inputs = (stoch, deter)
        tape.state_planes = (stoch_p, deter_p)        # rows h*N + n: the heads evaluated on the rollout take them as operands
        ctx.sp = sp
        ctx.bufs = (stoch, deter, logit, raws, eps, x_pre, g_pre, o_pre, st)
        ctx.nparams = len(actor_params)
        ctx.dims = (H, N, S, K, D, A, U)
        # (views, never the buffers themselves: ctx.bufs holds `deter` and `raws`, and an OUTPUT tensor kept on ctx is a reference cycle
        # through its grad_fn that Python's collector cannot see -- every eager iteration's rollout buffers, 1.4 GiB, stayed alive)
        return stoch.reshape(H + 1, N, S, K), deter.view(H + 1, N, D), logit.reshape(H + 1, N, S, K), action[:, :, :A], raws.view(H, N, 2 * A)

    @staticmethod
    def backward(ctx, d_stoch, d_deter, d_logit, d_action, d_raws):
        sp, tape = ctx.sp, ctx.sp.tape
        stoch, deter, logit, raws, eps, x_pre, g_pre, o_pre, st = ctx.bufs
        H, N, S, K, D, A, U = ctx.dims
        SK = S * K
        dev = deter.device
        AP = (A + 3) // 4 * 4
        z = lambda *shape: torch.zeros(*shape, device=dev)
        f = lambda *shape: torch.empty(*shape, device=dev)
        ds = d_stoch.reshape(H + 1, N, SK).clone() if d_stoch is not None else z(H + 1, N, SK)
        dd = d_deter.clone() if d_deter is not None else z(H + 1, N, D)
        dl_in = d_logit.reshape(H + 1, N, SK).contiguous() if d_logit is not None else None
        da_in = d_action.contiguous() if d_action is not None else None
        # (d o_pre is consumed as planes only: the LayerNorm backward writes no fp32 copy of it where its kernel can do without)
        no_dx = 256 < U <= 4096 and U % 4 == 0
        dlg, do, do_pre, dg_pre, dx, dx_pre = f(N, SK), f(N, U), (None if no_dx else f(N, U)), f(N, 3 * D), f(N, U), f(N, U)
        dlg_p, dop_p, dg_p, dxp_p = planes.Planes(N, SK, dev), planes.Planes(N, U, dev), planes.Planes(N, 3 * D, dev), planes.Planes(N, U, dev)
        dha, dhb = f(N, D), f(N, D)
        cur, nxt = dha, None                              # ping-pong: recurrent gradient into deter_h from step h's GRU
        # transposed weight planes: rows = the product's output columns
        wt_dist, wt_out = planes.weight(sp.dist_w, True), planes.weight(sp.out_w, True)
        wt_g_x, wt_g_h = planes.weight(sp.gru_w, True, 0, U), planes.weight(sp.gru_w, True, U)
        wt_in_s = planes.weight(sp.in_w, True, 0, SK)
        waT = sp.in_w.detach()[:, SK:SK + A].t().contiguous()            # (A, U): the action columns, for the fused head backward
        pt = lambda t, off: t.data_ptr() + 4 * off
        L = lib()
        dact_all = None
        if da_in is not None:                 # upstream action gradients, once, in rows padded like the forward's actions
            dact_all = torch.zeros(H + 1, N, AP, device=dev)
            dact_all[:, :, :A].copy_(da_in)
        seq_c = ops.SEQ_C and planes.gemm_profile is None
        if seq_c:
            # the dgrad chain's launch loop in C (csrc/seq.hip: genrl_imagine_seq_bwd -- the loop below, launch for launch)
            a = _RolloutBwdArgs()
            a.H, a.N, a.S, a.K, a.D, a.A, a.AP, a.U = H, N, S, K, D, A, AP, U
            a.unimix, a.min_std, a.max_std = UNIMIX, sp.min_std, sp.max_std
            for n_, t_ in (('logit', logit), ('deter', deter), ('raws', raws), ('eps', eps), ('x_pre', x_pre), ('g_pre', g_pre), ('o_pre', o_pre),
                           ('xm', st['xm']), ('xr', st['xr']), ('gm', st['gm']), ('gr', st['gr']), ('om', st['om']), ('orr', st['or']),
                           ('ds', ds), ('dd', dd), ('dl_in', dl_in), ('dact_all', dact_all), ('d_raw', tape.d_raw), ('dlg', dlg), ('dov', do),
                           ('do_pre', do_pre), ('dg_pre', dg_pre), ('dx', dx), ('dx_pre', dx_pre), ('dha', dha), ('dhb', dhb), ('waT', waT),
                           ('out_g', sp.out_g), ('out_be', sp.out_be), ('gru_g', sp.gru_g), ('gru_be', sp.gru_be), ('in_g', sp.in_g),
                           ('in_be', sp.in_be)):
                setattr(a, n_, _p(t_))
            for n_, P_ in (('dlg_p', dlg_p), ('dop_p', dop_p), ('dg_p', dg_p), ('dxp_p', dxp_p), ('wt_dist', wt_dist), ('wt_out', wt_out),
                           ('wt_g_x', wt_g_x), ('wt_g_h', wt_g_h), ('wt_in_s', wt_in_s)):
                setattr(a, n_, _pref(P_))
            check(L.genrl_imagine_seq_bwd(_ct.addressof(a), _stream()), 'imagine_seq_bwd')
        for h in (() if seq_c else range(H - 1, -1, -1)):
            r0, r1 = h * N, (h + 1) * N
            if dl_in is not None:
                dlg.copy_(dl_in[h + 1])
            check(L.genrl_onehot_bwd_h2(pt(logit, r1 * SK), pt(ds, r1 * SK), _p(dlg), N * S, K, UNIMIX, int(dl_in is not None),
                                        dlg_p.ptr(), SK, dlg_p.ld, dlg_p.plane, dlg_p.inv_ptr(), _stream()), 'onehot_bwd_h2')
            planes.gemm(dlg_p, wt_dist, do, U, None, N, U)
            _ln_bwd(_p(do), pt(o_pre, h * N * U), sp.out_g, sp.out_be, pt(st['om'], r0), pt(st['or'], r0), _p(do_pre), N, U,
                    dop_p, 0)
            planes.gemm(dop_p, wt_out, dd, D, None, N, D, accumulate=True, c_off=r1 * D)
            # GRU: upstream = dd[h+1] (+ recurrent part from step h+1's GRU, held in `nxt`)
            check(L.genrl_gru_gates_bwd_h2(pt(dd, r1 * D), D, nxt.data_ptr() if nxt is not None else None, None,
                                           pt(g_pre, h * N * 3 * D), pt(deter, r0 * D), D, _p(sp.gru_g), _p(sp.gru_be),
                                           pt(st['gm'], r0), pt(st['gr'], r0), _p(dg_pre), _p(cur), D, None, None, None, N, D,
                                           0, None, 0, 0, dg_p.ptr(), dg_p.ld, dg_p.plane, dg_p.inv_ptr(), _stream()),
                  'gru_gates_bwd_h2')
            planes.gemm(dg_p, wt_g_h, cur, D, None, N, D, accumulate=True)
            planes.gemm(dg_p, wt_g_x, dx, U, None, N, U)
            _ln_bwd(_p(dx), pt(x_pre, h * N * U), sp.in_g, sp.in_be, pt(st['xm'], r0), pt(st['xr'], r0), _p(dx_pre), N, U,
                    dxp_p, 0)
            planes.gemm(dxp_p, wt_in_s, ds, SK, None, N, SK, accumulate=True, c_off=r0 * SK)
            # d action_{h+1} = dx_pre W_a (+ upstream) and the head's backward -> d raw_h: one launch
            check(L.genrl_actor_head_linear_bwd(_p(dx_pre), U, _p(waT), pt(dact_all, r1 * AP) if dact_all is not None else None, AP,
                                                pt(raws, h * N * 2 * A), pt(eps, h * N * A), pt(tape.d_raw, h * N * 2 * A), N, U, A,
                                                sp.min_std, sp.max_std, _stream()), 'actor_head_linear_bwd')
            nxt, cur = cur, (dhb if cur is dha else dha)
        if d_raws is not None:
            tape.d_raw += d_raws
        dWh, dbh, grads = tape._backward()
        flat = [g for lg in grads for g in lg]
        return (None, None, None, None, None, None, dWh, dbh, *flat)


class _DenseLNActPlanes(Function):
    """ops._DenseLNAct (y = SiLU(LayerNorm([x1, x2] W^T + b)), agent/dreamer_utils.py:739-747) with plane operands: the forward
    product runs on planes when the inputs come with them (P1 / P2: Planes handles + first row), the output's planes are written by
    the LayerNorm kernel (out_p), and the backward's dgrad products use the planes its LayerNorm backward emits and the
    transposed weight planes.  Weight gradients: fp32-operand kernels, as everywhere."""
    @staticmethod
    def forward(ctx, x1, x2, W, b, gamma, beta, eps, P1, r1, P2, r2, out_p):
        a = _f32(x1).reshape(-1, x1.shape[-1]).contiguous()
        c = _f32(x2).reshape(-1, x2.shape[-1]).contiguous() if x2 is not None else None
        M, K1 = a.shape
        K2 = c.shape[1] if c is not None else 0
        N, K = W.shape
        assert K == K1 + K2
        pre = torch.empty(M, N, device=a.device)
        y = torch.empty_like(pre)
        mean = torch.empty(M, device=a.device); rstd = torch.empty(M, device=a.device)
        on_planes = P1 is not None and (c is None or P2 is not None)
        if on_planes and planes.gemm_ln_ok(M, N, a.device):
            # product + LayerNorm + SiLU in ONE launch (row statistics exchanged inside one XCD's L2: genrl_gemm_h2_ln)
            if c is None:
                planes.gemm_ln(P1, planes.weight(W), pre, b, M, N, gamma, beta, eps, out_p, 0, y=y, mean=mean, rstd=rstd, a_row0=r1)
            else:
                planes.gemm_ln(P1, planes.weight(W, c0=0, c1=K1), pre, b, M, N, gamma, beta, eps, out_p, 0, y=y, mean=mean, rstd=rstd,
                               a_row0=r1, A1=P2, B1=planes.weight(W, c0=K1), a1_row0=r2)
        else:
            if on_planes:
                if c is None:
                    planes.gemm(P1, planes.weight(W), pre, N, b, M, N, a_row0=r1)
                else:
                    planes.gemm(P1, planes.weight(W, c0=0, c1=K1), pre, N, b, M, N, a_row0=r1, A1=P2, B1=planes.weight(W, c0=K1), a1_row0=r2)
            else:
                w1, ld1 = ops._aligned_block(W, K1, M)
                sgemm(a, K1, 1, w1, ld1, 1, pre, N, b, M, N, K1)
                if c is not None:
                    sgemm(c, K2, 1, W, K, 1, pre, N, None, M, N, K2, accumulate=True, b_off=K1)
            _ln_fwd(_p(pre), gamma, beta, _p(y), _p(mean), _p(rstd), M, N, eps, out_p, 0)
        ctx.save_for_backward(a, c if c is not None else a.new_empty(0), W, gamma, beta, pre, mean, rstd)
        ctx.has2 = c is not None
        ctx.bias = b
        # the inputs' planes serve the weight gradient again (genrl_gemm_h2_tn): kept with the node
        # (only when that product will really take them: otherwise the fp16 copies would live from forward to backward for nothing)
        keep = (P1 is not None and (c is None or P2 is not None) and ctx.needs_input_grad[2] and planes.tn_ok(M, N, K1, K)
                and (c is None or K2 % 4 == 0) and r1 % 4 == 0 and r2 % 4 == 0)
        ctx.in_planes = ((P1, r1), (P2, r2)) if keep else None
        ctx.shapes = (x1.shape, x2.shape if x2 is not None else None)
        return y.reshape(*x1.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        a, c, W, gamma, beta, pre, mean, rstd = ctx.saved_tensors
        M, K1 = a.shape
        K2 = c.shape[1] if ctx.has2 else 0
        N, K = W.shape
        dev = dy.device
        b = ctx.bias
        dy2 = dy.reshape(M, N).contiguous()
        ip = ctx.in_planes
        use_tn = ip is not None and planes.tn_ok(M, N, K1, K) and (not ctx.has2 or K2 % 4 == 0)
        # the fp32 copy of the pre-activation gradient has one reader, the fp32-operand weight-gradient kernel: with that product
        # on planes (genrl_gemm_h2_tn) the LayerNorm backward writes planes only (a quarter of its traffic less)
        planes_only = 256 < N <= 4096 and N % 4 == 0          # (the LayerNorm kernels that write planes themselves)
        dpre = torch.empty_like(pre) if ((ctx.needs_input_grad[2] and not use_tn) or not planes_only) else None
        dpre_p = planes.Planes(M, N, dev)
        need_p = ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        tg, tb, tc = _grad_buf(gamma), _grad_buf(beta), (_grad_buf(b) if b is not None else None)
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
            acc_p |= ops.defer_reduce(M, N, ws, g0, g1, g2)
        _ln_bwd(_p(dy2), _p(pre), gamma, beta, _p(mean), _p(rstd), _p(dpre) if dpre is not None else None, M, N, dpre_p, 0, g0, g1, g2, ws, acc_p)
        d1 = d2 = dW = None
        if ctx.needs_input_grad[0]:
            d1 = torch.empty(M, K1, device=dev)
            planes.gemm(dpre_p, planes.weight(W, True, 0, K1), d1, K1, None, M, K1)
            d1 = d1.reshape(ctx.shapes[0])
        if ctx.has2 and ctx.needs_input_grad[1]:
            d2 = torch.empty(M, K2, device=dev)
            planes.gemm(dpre_p, planes.weight(W, True, K1), d2, K2, None, M, K2)
            d2 = d2.reshape(ctx.shapes[1])
        if ctx.needs_input_grad[2]:
            tgt = _grad_buf(W)
            acc = tgt is not None
            if not acc:
                dW = tgt = torch.empty(N, K, device=dev)
            if use_tn:
                planes.gemm_tn(dpre_p, ip[0][0], tgt, K, N, K1, M, accumulate=acc, b_row0=ip[0][1])
                if ctx.has2:
                    planes.gemm_tn(dpre_p, ip[1][0], tgt, K, N, K2, M, accumulate=acc, b_row0=ip[1][1], c_off=K1)
            else:
                sgemm(dpre, 1, N, a, 1, K1, tgt, K, None, N, K1, M, accumulate=acc)
                if ctx.has2:
                    sgemm(dpre, 1, N, c, 1, K2, tgt, K, None, N, K2, M, accumulate=acc, c_off=K1)
        if need_p and not direct:
            return d1, d2, dW, (g2 if b is not None else None), g0, g1, None, None, None, None, None, None
        return d1, d2, dW, None, None, None, None, None, None, None, None, None


class _LinearPlanes(Function):
    """ops._Linear (y = x W^T + b) with the forward and dgrad products on plane operands: P = (Planes of x, first row) when
    the caller has them (a rollout's states), else x is split here; the dgrad splits dy.  Weight / bias gradients: fp32-operand
    kernels (ops.sgemm / colsum), as everywhere."""
    @staticmethod
    def forward(ctx, x, W, b, P, r0):
        x2 = _f32(x).reshape(-1, x.shape[-1]).contiguous()
        M, K = x2.shape
        N = W.shape[0]
        if P is None:
            P, r0 = planes.split(x2), 0
        Np = (N + 3) // 4 * 4            # (255-bin two-hot heads: rows padded to 256 floats, the caller gets a column slice)
        y = torch.empty(M, Np, device=x.device)
        planes.gemm(P, planes.weight(W), y, Np, b, M, N, a_row0=r0)
        ctx.save_for_backward(x2, W)
        ctx.bias = b
        ctx.xshape = x.shape
        ctx.in_planes = (P, r0) if (ctx.needs_input_grad[1] and planes.tn_ok(M, N, K, K) and r0 % 4 == 0) else None
        return (y if Np == N else y[:, :N]).view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        M, K = x2.shape
        N = W.shape[0]
        b = ctx.bias
        dy2, ldy = ops._rows_ld(dy.reshape(M, N))        # (a gradient arriving in padded rows is read in place)
        dx = dW = db = None
        tn = ctx.needs_input_grad[1] and ctx.in_planes is not None and planes.tn_ok(M, N, K, K)
        dyp = planes.split(dy2) if (ctx.needs_input_grad[0] or tn) else None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device)
            planes.gemm(dyp, planes.weight(W, transpose=True), dx, K, None, M, K)
            dx = dx.reshape(ctx.xshape)
        if ctx.needs_input_grad[1]:
            tgt = _grad_buf(W)
            acc = tgt is not None
            if not acc:
                dW = tgt = torch.empty(N, K, device=dy.device)
            if tn:
                planes.gemm_tn(dyp, ctx.in_planes[0], tgt, K, N, K, M, accumulate=acc, b_row0=ctx.in_planes[1])
            else:
                sgemm(dy2, 1, ldy, x2, 1, K, tgt, K, None, N, K, M, accumulate=acc)
        if b is not None and ctx.needs_input_grad[2]:
            tgt = _grad_buf(b)
            if tgt is not None:
                colsum(dy2, out=tgt, accumulate=True, ld=ldy)
            else:
                db = colsum(dy2, ld=ldy)
        return dx, dW, db, None, None


def linear(x, W, b=None, planes_of_x=None):
    """planes_of_x = (Planes handle, first row) of x's rows, if the caller has them"""
    P, r0 = planes_of_x if planes_of_x is not None else (None, 0)
    M = x.numel() // x.shape[-1]
    if P is not None and (P.cols != x.shape[-1] or r0 + M > P.rows):
        P, r0 = None, 0
    return _LinearPlanes.apply(x, W, b, P, r0)


# rows from which products run on plane operands.  Round 5 (profiles/r05_minrows_ab.txt, whole step, one box): at 384 / 400 rows the plane
# rollout's 16 + 10 launches per step beat the fp32-operand rollout's 20 + 14 by 15 % / 7 % (17.0 -> 14.5 ms at 12 sequences of c2,
# 13.9 -> 12.9 at the 8 x 50 per-rank batch of c3 under DP-8), at 256 rows by 8-12 % (c5 9.4 -> 8.5 ms, c2 at 8 sequences 13.6 -> 12.5).
# Round 6 takes the measured win (round-5 verdict item 2; profiles/r06_minrows_ab.txt, one box: c5 9.5 -> 8.5 ms, c2 at 8 sequences = 256 rows
# 13.1 -> 12.2, at 6 sequences = 192 rows 12.1 -> 11.4; at 4 sequences = 128 rows the fp32-operand kernels still win, 10.0 against 10.5): plane
# operands from 192 rows.  With them one of the 8 192 sampled latents of the 256-row full-width c5 case falls on the other side of a near-tie
# -- the flip rate (1.2e-4) the suite accepts for c3 / c4 (< 2e-3) -- and GENRL_PLANES_MIN_ROWS=320 (or any larger value) is the explicit
# switch back to the exact-fp32 operands at that size (tests/test_gpu_fullsize.py runs the c5 case both ways).
MIN_ROWS_PLANES = 192


def min_rows():
    """rows from which the plane-operand path is used (GENRL_PLANES_MIN_ROWS overrides: the parity tests run it at tiny sizes)"""
    import os
    return int(os.environ.get('GENRL_PLANES_MIN_ROWS', MIN_ROWS_PLANES))


def dense_ln_act(x1, x2, W, b, gamma, beta, eps=1e-5, planes=None):
    """-> y with y._planes = (planes of y, 0) for the next layer.  planes = ((P1, row0), (P2, row0) | None) of the inputs when the
    caller has them (rollout states); otherwise the inputs' own `_planes` attribute (a previous layer's output) is used."""
    M = x1.numel() // x1.shape[-1]
    N = W.shape[0]
    if planes is None:
        h1 = getattr(x1, '_planes', None)
        h2 = getattr(x2, '_planes', None) if x2 is not None else None
    else:
        h1, h2 = planes[0], (planes[1] if len(planes) > 1 else None)
    if h1 is not None and (h1[0].cols != x1.shape[-1] or h1[1] + M > h1[0].rows):
        h1 = None
    if x2 is not None and (h2 is None or h2[0].cols != x2.shape[-1] or h2[1] + M > h2[0].rows):
        h1 = h2 = None
    P1, r1 = h1 if h1 is not None else (None, 0)
    P2, r2 = h2 if h2 is not None else (None, 0)
    out_p = pl.Planes(M, N, x1.device)
    y = _DenseLNActPlanes.apply(x1, x2, W, b, gamma, beta, float(eps), P1, r1, P2, r2, out_p)
    y._planes = (out_p, 0)
    return y


def imagine_rollout(stoch0, deter0, logit0, eps, q, spec):
    tape = spec.tape
    flat = [qq for l in tape.layers for qq in l[:4]]
    return _RolloutPlanes.apply(stoch0, deter0, logit0, eps, q, spec, tape.head_w, tape.head_b, *flat)
