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
