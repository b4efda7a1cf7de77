"""Random-number sites of the hot path.  The reference draws noise implicitly
(torch.multinomial in OneHotDist.sample, Normal.rsample, randn_like); here every
site asks this module for an explicit noise tensor which the HIP kernels consume
(exponential-race argmax for categoricals, eps for Gaussians).  By default noise comes from the
device's torch generator; `inject()` replays given tensors per site so that runs are comparable
across devices bit-for-bit in their noise ("identical replay batches and seeds")."""
import contextlib
import torch

_injected = None
_static = None       # {(site, shape): tensor}: see static()


def draw(kind, site, shape, device):
    """kind: 'exp' (Exp(1)) or 'normal' (N(0,1)); site: string naming the RNG call site."""
    if _injected is not None and site in _injected:
        q = _injected[site]
        t = q.pop(0) if isinstance(q, list) else q
        assert tuple(t.shape) == tuple(shape), (site, tuple(t.shape), tuple(shape))
        return t.to(device=device, dtype=torch.float32).contiguous()
    if _static is not None:
        # one fixed tensor per (site, shape), generated on first use from a seed derived from the site name: every
        # step (eager or a replayed hipGraph, which re-reads the same address) consumes identical noise
        key = (site, tuple(shape))
        if key not in _static:
            import zlib
            g = torch.Generator().manual_seed(_static_seed + zlib.crc32(site.encode()))
            rank, world = _static_dp
            dim = 1 if site in _ROWS_DIM1 else 0           # the dimension that enumerates the batch rows (b-major)
            full = list(shape)
            full[dim] *= world                               # the GLOBAL tensor; this rank consumes its row block
            t = torch.empty(tuple(full), dtype=torch.float32)
            t = t.exponential_(1.0, generator=g) if kind == 'exp' else t.normal_(generator=g)
            _static[key] = t.narrow(dim, rank * shape[dim], shape[dim]).contiguous().to(device)
        return _static[key]
    return _arena_draw(kind, shape, device)


# ---- one RNG launch per kind and iteration -------------------------------------------------------------------------
# An iteration makes ~12 draws (posterior / prior / imagination categoricals, action eps, connector noise): their sizes
# repeat from one iteration to the next, so new_step() (called where an iteration starts: update_wm) draws ONE flat
# Exp(1) tensor and ONE flat N(0,1) tensor with the sizes the previous iteration asked for, and draw() hands out
# 256-byte aligned slices in call order.  Any draw that does not match the plan (first iteration, another call pattern,
# another device) falls back to its own launch and the plan is rebuilt from what was actually asked for.
_plan = {'exp': [], 'normal': []}        # numels of the previous iteration's draws, in order
_seen = {'exp': [], 'normal': []}        # ... of the running iteration
_flat = {'exp': None, 'normal': None}    # (flat tensor, [offsets]) for the running iteration
ARENA = True


def _pad(n):
    return (n + 63) // 64 * 64


def new_step(device):
    """start of an iteration: adopt the finished iteration's draw sizes as the plan and pre-draw the flat tensors"""
    if not ARENA or _injected is not None or _static is not None:
        return
    for kind in ('exp', 'normal'):
        if _seen[kind]:
            _plan[kind] = _seen[kind]
        _seen[kind] = []
        sizes = _plan[kind]
        if not sizes:
            _flat[kind] = None
            continue
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot); tot += _pad(n)
        t = torch.empty(tot, device=device, dtype=torch.float32)
        t = t.exponential_(1.0) if kind == 'exp' else t.normal_()
        _flat[kind] = (t, offs)


def _arena_draw(kind, shape, device):
    n = 1
    for d in shape:
        n *= int(d)
    i = len(_seen[kind])
    _seen[kind].append(n)
    fl = _flat[kind]
    d = torch.device(device)
    if (fl is not None and i < len(fl[1]) and _plan[kind][i] == n and fl[0].device.type == d.type
            and (d.index is None or d.index == fl[0].device.index)):
        return fl[0][fl[1][i]:fl[1][i] + n].view(tuple(shape))
    if kind == 'exp':
        return torch.empty(shape, device=device, dtype=torch.float32).exponential_(1.0)
    return torch.randn(shape, device=device, dtype=torch.float32)


@contextlib.contextmanager
def inject(sites):
    """sites: dict site -> tensor, or -> list of tensors consumed first-in-first-out."""
    global _injected
    prev = _injected
    _injected = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in sites.items()}
    try:
        yield
    finally:
        _injected = prev


_static_seed = 0
_static_dp = (0, 1)
# sites whose tensors are time-major (T or H first): their rows (b-major) sit in dimension 1; everywhere else in 0
_ROWS_DIM1 = {'wm.post_q', 'wm.prior_q', 'imag.act_eps', 'imag.step_q', 'rssm.imagine_q'}


@contextlib.contextmanager
def static(seed=0, cache=None, dp=(0, 1)):
    """Every draw returns a FIXED per-(site, shape) tensor (see draw): makes an eager iteration and a hipGraph
    replay of it -- or two processes -- consume bit-identical noise.  `cache`: share the tensors between contexts.
    dp = (rank, world): the tensor is the rank's row block of the noise a single process would draw for the
    world-times larger batch (DP-k == DP-1 needs row-sliced noise)."""
    global _static, _static_seed, _static_dp
    prev = (_static, _static_seed, _static_dp)
    _static, _static_seed, _static_dp = ({} if cache is None else cache), seed, tuple(dp)
    try:
        yield _static
    finally:
        _static, _static_seed, _static_dp = prev
