"""Side HIP streams for work that is independent of the main chain of an iteration.

The connector (VideoSSM) updates are ~600 tiny, strictly sequential kernels each (a T-step GRU scan
at batch 32: M=32 GEMMs) that leave the chip almost idle, and nothing in the imagination /
actor-critic phase reads or writes the connector (its rollout target is cached after the first
call, tools/genrl_utils.py:289-321).  Running them on a second stream lets their launch-latency-
bound chain hide behind the MFMA-bound imagination kernels.  `fork()` orders the side stream after
everything already enqueued on the current stream; `join()` makes the current stream wait for the
side work (called before the iteration's last API call returns, so callers observe ordinary
single-stream semantics).  Under hipGraph capture the fork/join becomes two graph branches."""
import contextlib
import torch

_side = {}
_dirty = set()


def _resolve_raw_stream():
    """-> callable returning the raw hipStream_t of torch's current stream on the current device.  torch._C._cuda_getCurrentRawStream is
    the C call behind torch.cuda.current_stream() (that wrapper costs ~8 us of host time per call and an eager iteration asks ~300 times);
    it is a private name, so a torch build without it gets the public call instead (same handle, slower)."""
    get_raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    get_dev = getattr(torch._C, '_cuda_getDevice', None)
    if callable(get_raw) and callable(get_dev):
        try:
            get_raw(get_dev())
            return lambda: get_raw(get_dev())
        except Exception:           # signature changed: fall through to the public API
            pass
    return lambda: torch.cuda.current_stream().cuda_stream


_raw = None


def raw_current_stream():
    """raw handle of torch's current stream (no CPU fallback: raises without a GPU).  Availability is asked for on every call until it
    holds (a device that appears later is noticed), the resolved getter is kept."""
    global _raw
    if _raw is None:
        if not torch.cuda.is_available():
            from ._lib import GenrlHipError
            raise GenrlHipError('genrl_amd ops need an MI355X (torch.cuda unavailable); there is no CPU fallback')
        torch.cuda.init()            # (the raw call skips torch.cuda's lazy initialisation)
        _raw = _resolve_raw_stream()
    return _raw()


def _stream(name):
    dev = torch.cuda.current_device()
    key = (name, dev)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=dev)
    return key, _side[key]


@contextlib.contextmanager
def fork(name):
    key, s = _stream(name)
    s.wait_stream(torch.cuda.current_stream())
    _dirty.add(key)
    with torch.cuda.stream(s):
        yield s


def join(name=None):
    cur = torch.cuda.current_stream()
    for key in list(_dirty):
        if key[1] == torch.cuda.current_device() and (name is None or key[0] == name):
            cur.wait_stream(_side[key])
            _dirty.discard(key)


def pending(name):
    """has work been forked onto side stream `name` (on the current device) that nobody has joined yet?"""
    return (name, torch.cuda.current_device()) in _dirty


def on_side_stream():
    """is torch's current stream one of the side streams created here (on the current device)?"""
    cur = raw_current_stream()
    dev = torch.cuda.current_device()
    return any(k[1] == dev and s.cuda_stream == cur for k, s in _side.items())
