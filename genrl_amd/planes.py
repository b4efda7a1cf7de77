"""h2 planes: host-side handles of the pre-split GEMM operands of csrc/gemm_planes.hip.

An fp32 matrix (rows x cols) is held as TWO fp16 planes of its row-scaled values, a s = h + l / 2^11 (s a power of two
per row: the row's largest magnitude lands in [2^14, 2^15)), in ONE int16 tensor [2, rows, ld] with ld = cols rounded up
to 64 (zero padded), plus inv[rows] = 1 / s: the operand format of genrl_gemm_h2, which does fp32-accurate products on
the fp16 matrix cores (three MFMAs per block and k-step) with no conversion work in its K loop.  Activations get their
planes from the producing row kernel (ops: *_h2 entry points); weights are split here, once per optimiser step
(`weight`, cached until `invalidate()`), also transposed for the dgrad products.  (The first plane format, three bf16 planes with six products --
"x3" -- is still offered by the kernel, genrl_split_x3 / genrl_gemm_x3, as the exactly-representing variant.)"""
import ctypes
import os
import weakref
import torch
from ._lib import lib, check, GenrlHipError

ENABLED = os.environ.get('GENRL_PLANES', '1') != '0'
_amp_saved = None            # ENABLED as it was before a precision-16 agent switched the plane products off (agent/dreamer.py)
gemm_profile = None          # bench.py: list of (M, N, K, start_event, end_event, tag)


from .streams import raw_current_stream as _stream      # raw hipStream_t of torch's current stream (private fast call, public fallback)


def r64(k):
    return (k + 63) // 64 * 64


class Planes:
    """planes of a (rows x cols) matrix; .t int16 [2, rows, ld] (fp16 bits), .inv fp32 [rows]"""
    __slots__ = ('t', 'inv', 'rows', 'cols', 'ld', 'plane', 'uniform')

    def __init__(self, rows, cols, dev, zero=True):
        self.rows, self.cols, self.ld = rows, cols, r64(cols)
        self.uniform = False         # ONE scale for the whole tensor (the convolution operands: set by their producers)
        # padding columns must hold zeros (0 x garbage may be NaN): zero-filled once when there are any -- unless the
        # producer writes the padding itself (zero=False: the channel-LayerNorm and uniform-split kernels do)
        mk = torch.zeros if (self.ld != cols and zero) else torch.empty
        self.t = mk(2, rows, self.ld, dtype=torch.int16, device=dev)
        self.inv = torch.empty(rows, device=dev)
        self.plane = rows * self.ld

    def ptr(self, row0=0):
        return self.t.data_ptr() + 2 * row0 * self.ld

    def inv_ptr(self, row0=0):
        return self.inv.data_ptr() + 4 * row0

    def float(self):
        """back to fp32 (tests): (h + l / 2^11) * inv, evaluated in float64"""
        f = lambda t: t.view(torch.float16).double()
        return ((f(self.t[0]) + f(self.t[1]) / 2048.0) * self.inv.double()[:, None])[:, :self.cols].float()


def split(x2d, transpose=False, out=None, row0=0):
    """planes of the 2-D fp32 tensor x2d (rows evenly spaced, unit column stride) or of its transpose; `out`/`row0`:
    write into rows row0.. of an existing handle"""
    assert x2d.dim() == 2 and x2d.dtype == torch.float32 and (x2d.shape[1] == 1 or x2d.stride(1) == 1)
    R, C = x2d.shape
    Ro, Co = (C, R) if transpose else (R, C)
    if out is None:
        out = Planes(Ro, Co, x2d.device)
    assert out.cols == Co and row0 + Ro <= out.rows
    check(lib().genrl_split_h2(x2d.data_ptr(), x2d.stride(0), R, C, out.ptr(row0), out.ld, out.plane, out.inv_ptr(row0),
                               int(transpose), _stream()), 'split_h2')
    return out


# ---- weights: split once per optimiser step -------------------------------------------------------------------------
_epoch = 0
_wcache = {}             # (id(W), transpose, c0, c1) -> [epoch of the split, Planes, W.data_ptr(), weakref(W), stream of the last use]
#                          (weak: an agent that goes away takes its cached planes along; the finalizer drops the entry, so a
#                          recycled id() can never meet a stale one)


class _Desc(ctypes.Structure):          # genrl_split_desc (include/genrl_hip.h)
    _fields_ = [('src', ctypes.c_void_p), ('ldx', ctypes.c_long), ('R', ctypes.c_int), ('Cn', ctypes.c_int),
                ('out', ctypes.c_void_p), ('ld_out', ctypes.c_long), ('plane', ctypes.c_long), ('inv', ctypes.c_void_p),
                ('transpose', ctypes.c_int)]


_dcache = {}             # (id(W), tag) -> [epoch, object derived from W, W.data_ptr(), weakref(W), stream that built it, (W._version, extra)]


def derived(W, tag, build, extra=None):
    """build() -- something computed from the parameter W alone (a permuted copy, planes of a rearrangement) -- cached until W is
    invalidated (optimiser step, slow-target copy, load_state_dict: `invalidate`), rebuilt at the first use after that on the stream
    that uses it; a use on another stream than the one that built it rebuilds too (no cross-stream ordering is assumed).
    Inside a captured iteration everything is stale at the capture's start, so every replay rebuilds at the same place.
    (Round 5's rebuild-ahead-of-use on a side stream, `prefetch`, measured slower -- profiles/r05_prefetch_ab.txt -- and is gone.)"""
    if not isinstance(W, torch.nn.Parameter):
        return build()
    key, st = (id(W), tag), _stream()
    ent = _dcache.get(key)
    # (W._version catches in-place edits made through torch ON THE PARAMETER ITSELF -- W.copy_ / nn.init under no_grad, load_state_dict.  It
    # does NOT see edits through `.data` (its own version counter), through views of the optimiser's flat buffers (a broadcast of weights
    # into g.flat, an EMA copy) or through raw pointers (the optimiser's kernels): those callers invalidate explicitly -- Optimizer.__call__,
    # the slow-target copy, bench.resync_weights.  `extra`: a version of something else the object is built from -- the bias of a sub-pixel
    # weight -- checked here so that a stale entry is REPLACED, not kept beside a new one under another key.)
    ver = (W._version, extra)
    fresh = ent is not None and ent[0] == _epoch and ent[2] == W.data_ptr() and ent[3]() is W and ent[5] == ver
    if not fresh or ent[4] != st:
        if ent is None:
            weakref.finalize(W, _dcache.pop, key, None)
        ent = _dcache[key] = [_epoch, build(), W.data_ptr(), weakref.ref(W), st, ver]
    return ent[1]


def invalidate(params=None):
    """some parameters changed (optimiser step, slow-target copy, load_state_dict): their cached planes are stale.
    params: the tensors that changed (None: everything)"""
    global _epoch
    if params is None:
        _epoch += 1
        return
    ids = {id(q) for q in params}
    for key, ent in _wcache.items():
        if key[0] in ids:
            ent[0] = -1
    for key, ent in _dcache.items():
        if key[0] in ids:
            ent[0] = -1


def _split_entries(stale, stream):
    if not stale:
        return
    arr = (_Desc * len(stale))()
    for d, ((_, transpose, c0, c1), ent) in zip(arr, stale):
        W, P = ent[3](), ent[1]
        d.src, d.ldx, d.R, d.Cn = W.data_ptr() + 4 * c0 * W.stride(1), W.stride(0), W.shape[0], c1 - c0
        d.out, d.ld_out, d.plane, d.inv, d.transpose = P.ptr(0), P.ld, P.plane, P.inv_ptr(0), int(transpose)
    check(lib().genrl_split_h2_batch(ctypes.cast(arr, ctypes.c_void_p), len(stale), stream), 'split_h2_batch')
    for _, ent in stale:
        ent[0], ent[2] = _epoch, ent[3]().data_ptr()


def _refresh_stale(stream):
    """re-split every stale cached weight LAST USED ON THIS STREAM in one batched launch set (genrl_split_h2_batch): after an
    optimiser step all of a group's matrices are stale together, and the first one asked for brings the others along.
    (Weights used on another stream -- the connector's on the side stream -- are left to that stream: a refresh enqueued here
    would not be ordered before their use there.)"""
    _split_entries([(key, ent) for key, ent in _wcache.items() if ent[0] != _epoch and ent[4] == stream and ent[3]() is not None], stream)


def refresh(params):
    """re-split the cached planes of these parameters NOW, on the current stream (a change made outside a captured iteration
    -- the slow-critic copy between two graph replays -- cannot wait for a refresh that the captured graph may not contain)"""
    ids = {id(q) for q in params}
    _split_entries([(key, ent) for key, ent in _wcache.items() if key[0] in ids and ent[3]() is not None], _stream())


def weight(W, transpose=False, c0=0, c1=None):
    """planes of W[:, c0:c1] (2-D) or of its transpose.  Cached for nn.Parameters until they are invalidated -- inside a
    captured iteration the refresh sits wherever the capture-time staleness put it, i.e. after the optimiser step that
    precedes the first use, as in every eager iteration."""
    c1 = W.shape[1] if c1 is None else c1
    if not isinstance(W, torch.nn.Parameter):
        return split(W.detach()[:, c0:c1], transpose)
    key = (id(W), transpose, c0, c1)
    ent = _wcache.get(key)
    st = _stream()
    if ent is None or ent[2] != W.data_ptr() or ent[3]() is not W:
        rows, cols = (c1 - c0, W.shape[0]) if transpose else (W.shape[0], c1 - c0)
        assert W.dim() == 2 and W.stride(1) == 1
        if ent is None:
            weakref.finalize(W, _wcache.pop, key, None)
        ent = _wcache[key] = [-1, Planes(rows, cols, W.device), W.data_ptr(), weakref.ref(W), st]
    ent[4] = st                       # the stream this weight is used on
    if ent[0] != _epoch:
        _refresh_stale(st)
    return ent[1]


def gemm(A, B, C, ldc, bias, M, N, accumulate=False, a_row0=0, A1=None, B1=None, a1_row0=0, c_off=0, b_row0=0, b1_row0=0):
    """C[M, N] (+)= A[a_row0.., :] B^T (+ A1 B1^T) (+ bias); A, B: Planes handles; C fp32 tensor, c_off elements in"""
    assert A.ld == B.ld and a_row0 + M <= A.rows and b_row0 + N <= B.rows, (A.ld, B.ld, A.rows, B.rows, M, N)
    if gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    k1 = 0
    a1 = b1 = (None, 0, 0, None)
    if A1 is not None:
        assert A1.ld == B1.ld and a1_row0 + M <= A1.rows and b1_row0 + N <= B1.rows
        k1 = A1.ld
        a1, b1 = (A1.ptr(a1_row0), A1.ld, A1.plane, A1.inv_ptr(a1_row0)), (B1.ptr(b1_row0), B1.ld, B1.plane, B1.inv_ptr(b1_row0))
    check(lib().genrl_gemm_h2(A.ptr(a_row0), A.ld, A.plane, A.inv_ptr(a_row0), B.ptr(b_row0), B.ld, B.plane, B.inv_ptr(b_row0),
                              A.ld, *a1, *b1, k1, C.data_ptr() + 4 * c_off, ldc, bias.data_ptr() if bias is not None else None,
                              M, N, int(accumulate), _stream()), 'gemm_h2')
    if gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        gemm_profile.append((M, N, A.cols + (A1.cols if A1 is not None else 0), e0, e1, 'kk/h2/pipe4'))


# ---- Dense -> LayerNorm (-> SiLU) in one launch (genrl_gemm_h2_ln: the column tiles of a row block exchange their row statistics inside ONE
# XCD's L2 behind a barrier of N / 64 workgroups, csrc/gemm_planes.hip LnEpi).  Two such launches must never run concurrently (each would wait
# for workgroups the other keeps off the CUs), so the fused form is taken on the iteration's MAIN stream only -- never on a side stream of
# genrl_amd/streams.py -- and its workspace (barrier counters, zeroed once; exchange slab) is one per device, created outside graph capture.
LN_FUSED = os.environ.get('GENRL_GEMM_LN', '1') != '0'
_ln_ws = {}              # device index -> (failure word (int32), exchange records (fp32; zeroed once: the records carry launch-count tags))


def _ln_workspace(dev):
    d = torch.device(dev).index
    d = torch.cuda.current_device() if d is None else d
    ws = _ln_ws.get(d)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None              # (first use inside a capture: that call stays unfused; the warm-up iterations create it)
        L = lib()
        ws = _ln_ws[d] = (torch.zeros(L.genrl_gemm_h2_ln_sync_words(), dtype=torch.int32, device=f'cuda:{d}'),
                          torch.zeros(L.genrl_gemm_h2_ln_part_floats(1024, 1024) + 4, device=f'cuda:{d}'))
    return ws


def ln_fused_here(dev=None):
    """the shape-independent half of gemm_ln_ok: switched on, not on a side stream, workspace present"""
    if not (LN_FUSED and ENABLED):
        return False
    from . import streams
    return not streams.on_side_stream() and _ln_workspace(dev if dev is not None else torch.cuda.current_device()) is not None


def gemm_ln_ok(M, N, dev=None):
    """may y = act(LayerNorm(A B^T + b)) run as ONE launch here?  (shape: N % 64 == 0, N <= 1024, every workgroup resident at once; stream: not
    one of the side streams; workspace present)"""
    return LN_FUSED and ENABLED and lib().genrl_gemm_h2_ln_ok(M, N) == 1 and ln_fused_here(dev)


def gemm_ln(A, B, C, bias, M, N, gamma, beta, eps, out_p, out_row0, y=None, mean=None, rstd=None, act=True, a_row0=0, A1=None, B1=None,
            a1_row0=0, c_off=0, y_off=0, m_off=0):
    """C[M, N] = A[a_row0.., :] B^T (+ A1 B1^T) + bias (row stride N) and, in the same launch, y = act(LayerNorm(C) gamma + beta): fp32 rows
    into `y` (row stride N, y_off elements in; None: planes only), planes into rows out_row0.. of out_p (uniform scale), mean / rstd (m_off in)"""
    assert A.ld == B.ld and a_row0 + M <= A.rows and N <= B.rows and out_p.cols == N and out_row0 + M <= out_p.rows
    if gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    k1 = 0
    a1 = b1 = (None, 0, 0, None)
    if A1 is not None:
        assert A1.ld == B1.ld and a1_row0 + M <= A1.rows and N <= B1.rows
        k1 = A1.ld
        a1, b1 = (A1.ptr(a1_row0), A1.ld, A1.plane, A1.inv_ptr(a1_row0)), (B1.ptr(0), B1.ld, B1.plane, B1.inv_ptr(0))
    cptr = C.data_ptr() + 4 * c_off
    yptr = (y.data_ptr() + 4 * y_off) if y is not None else 0
    if (cptr | yptr | gamma.data_ptr() | beta.data_ptr() | (bias.data_ptr() if bias is not None else 0)) & 15:
        # (a parameter or an output row that is not 16-byte aligned -- not the case for the optimiser's flat buffers and torch's own
        # allocations: the two launches take any alignment)
        gemm(A, B, C, N, bias, M, N, a_row0=a_row0, A1=A1, B1=B1, a1_row0=a1_row0, c_off=c_off)
        mean_ = mean if mean is not None else torch.empty(M, device=C.device)
        rstd_ = rstd if rstd is not None else torch.empty(M, device=C.device)
        mo = m_off if mean is not None else 0
        check(lib().genrl_ln_act_fwd_h2(cptr, N, gamma.data_ptr(), beta.data_ptr(), yptr or None, N, mean_.data_ptr() + 4 * mo,
                                        rstd_.data_ptr() + 4 * (m_off if rstd is not None else 0), M, N, float(eps), int(act),
                                        out_p.ptr(out_row0), out_p.ld, out_p.plane, out_p.inv_ptr(out_row0), _stream()), 'ln_act_fwd_h2')
        return
    sync, part = _ln_workspace(C.device)
    check(lib().genrl_gemm_h2_ln(A.ptr(a_row0), A.ld, A.plane, A.inv_ptr(a_row0), B.ptr(0), B.ld, B.plane, B.inv_ptr(0), A.ld, *a1, *b1, k1,
                                 C.data_ptr() + 4 * c_off, N, bias.data_ptr() if bias is not None else None, M, N,
                                 gamma.data_ptr(), beta.data_ptr(), float(eps), int(act),
                                 (y.data_ptr() + 4 * y_off) if y is not None else None, N,
                                 (mean.data_ptr() + 4 * m_off) if mean is not None else None,
                                 (rstd.data_ptr() + 4 * m_off) if rstd is not None else None,
                                 out_p.ptr(out_row0), out_p.ld, out_p.plane, out_p.inv_ptr(out_row0),
                                 (part.data_ptr() + 15) // 16 * 16, sync.data_ptr(), _stream()), 'gemm_h2_ln')
    if gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        gemm_profile.append((M, N, A.cols + (A1.cols if A1 is not None else 0), e0, e1, 'kk/h2ln/pipe4'))


def check_ln_failure():
    """(synchronises) did a fused Dense -> LayerNorm launch raise its failure word -- a barrier that timed out (its workgroups were not
    co-resident) or a workgroup placement other than b % 8?  Turns the fused form off and raises: results since the last check are invalid."""
    global LN_FUSED
    for d, (sync, part) in _ln_ws.items():
        if int(sync[0].item()) != 0:
            LN_FUSED = False
            sync.zero_(); part.zero_()
            raise GenrlHipError('genrl_gemm_h2_ln: the exchange of row statistics inside an XCD timed out (workgroups not dealt round-robin over the XCDs, or not co-resident); '
                                'the fused Dense -> LayerNorm form is now off (GENRL_GEMM_LN=0 turns it off from the start)')


def gemm_sample(A, B, C, ldc, bias, M, N, q, ldq, unimix, sample, lds, SP=None, a_row0=0, c_off=0, q_off=0, s_off=0, sp_row0=0):
    """C[M, N] = A[a_row0.., :] B^T + bias AND, in the same launch, the categorical sample of every 32-class latent of the
    rows (genrl_gemm_h2_sample): one-hot rows into `sample` (fp32, s_off elements in) and their planes into SP (rows sp_row0..)"""
    assert A.ld == B.ld and a_row0 + M <= A.rows and N <= B.rows and N % 32 == 0
    if gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    check(lib().genrl_gemm_h2_sample(A.ptr(a_row0), A.ld, A.plane, A.inv_ptr(a_row0), B.ptr(0), B.ld, B.plane, B.inv_ptr(0), A.ld,
                                     C.data_ptr() + 4 * c_off, ldc, bias.data_ptr() if bias is not None else None, M, N,
                                     q.data_ptr() + 4 * q_off, ldq, float(unimix), sample.data_ptr() + 4 * s_off, lds,
                                     SP.ptr(sp_row0) if SP is not None else None, SP.ld if SP is not None else 0,
                                     SP.plane if SP is not None else 0, SP.inv_ptr(sp_row0) if SP is not None else None, _stream()),
          'gemm_h2_sample')
    if gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        gemm_profile.append((M, N, A.cols, e0, e1, 'kk/h2/pipe4'))


def gemm_tn(A, B, C, ldc, NI, NJ, M, accumulate=False, a_row0=0, b_row0=0, c_off=0):
    """C[NI, NJ] (+)= A[a_row0 .. a_row0 + M, :NI]^T B[b_row0 .. b_row0 + M, :NJ]: the weight gradient dW = dY^T X on the planes
    the row kernels emit for the forward / dgrad products (genrl_gemm_h2_tn: transposing LDS reads, the row scales folded
    into the fragments).  A, B: Planes handles with >= M rows from their first row; M % 64 == 0."""
    assert M % 64 == 0 and a_row0 + M <= A.rows and b_row0 + M <= B.rows and NI <= A.ld and NJ <= B.ld, (M, A.rows, B.rows)
    assert a_row0 % 4 == 0 and b_row0 % 4 == 0            # (the scales are read 16 bytes at a time)
    if gemm_profile is not None:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
    nb = lib().genrl_gemm_h2_tn_ws_bytes(NI, NJ, M)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=C.device)
    wp = (ws.data_ptr() + 255) // 256 * 256
    check(lib().genrl_gemm_h2_tn(A.ptr(a_row0), A.ld, A.plane, A.inv_ptr(a_row0), B.ptr(b_row0), B.ld, B.plane, B.inv_ptr(b_row0),
                                 C.data_ptr() + 4 * c_off, ldc, NI, NJ, M, int(accumulate), wp, nb, _stream()), 'gemm_h2_tn')
    if gemm_profile is not None:
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        gemm_profile.append((NI, NJ, M, e0, e1, 'rr/h2tn/pipe4'))


TN_ENABLED = os.environ.get('GENRL_PLANES_WGRAD', '1') != '0'


def tn_min_rows():
    """rows from which weight gradients take the plane kernel (below: few stages per workgroup, the fp32-operand kernels win;
    GENRL_TN_MIN_ROWS overrides -- the parity tests run it at tiny sizes)"""
    return int(os.environ.get('GENRL_TN_MIN_ROWS', '2048'))


def tn_ok(M, NI, NJ, ldc):
    """should genrl_gemm_h2_tn take this weight gradient?  (whole 64-row stages; 16-byte output rows for the split-K reduce)"""
    return ENABLED and TN_ENABLED and M % 64 == 0 and M >= tn_min_rows() and NJ % 4 == 0 and ldc % 4 == 0
