"""Replay-batch data parallelism: one process per GPU, `torch.distributed` (backend "nccl" is RCCL
over xGMI on ROCm).  The batch dimension B is sharded across ranks (sequences are independent,
SURVEY.md §8e); every optimiser group lives in ONE flat gradient buffer
(agent/dreamer_utils.FlatGroup), so a step issues exactly one sum-all-reduce per group
(world model, connector x2, actor, critic) right after its backward — the clip-norm and Adam
kernels then consume the reduced buffer with the 1/world factor folded in.  The world-model and actor
reductions are asynchronous (grad_reduce_async): they run beside the connector's / the critic's forward +
backward and are waited for just before their own clip / Adam pass.  With RCCL (stream-ordered collectives) the
connector updates -- their two reductions included -- stay on their side stream beside the imagination phase
(Optimizer.overlap_under_dp), and under hipGraph replay every collective is captured inside the one graph
(graph.GraphedStep(collectives='ingraph')): the only reduction nothing runs beside is the critic's, the last of the
iteration.  With gloo (tests) the connector stays on the main stream and three waits are exposed.  The RewardEMA
quantiles are taken over the all-gathered lambda-returns so every rank normalises identically.
Works with gloo on CPU tensors too (used by the world_size-2 tests)."""
import os
import torch
import torch.distributed as dist


def init(backend=None, ingraph=None):
    """Initialise from the torchrun environment.  Returns (rank, world, local_rank).

    ingraph: will collectives be captured inside hipGraphs (graph.GraphedStep(collectives='ingraph'))?  Default: the
    GENRL_DP_INGRAPH environment variable, else True for RCCL.  Only then is TORCH_NCCL_RETHROW_CUDA_ERRORS=0 set (see below);
    a run that keeps its collectives eager / at graph cuts (GENRL_DP_INGRAPH=0, bench.py --dp-graph cut) leaves the process
    group's default behaviour -- abort on an asynchronous device error -- in place."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = os.environ.get('GENRL_DP_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
            prepare_nccl_env(ingraph)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def prepare_nccl_env(ingraph=None):
    """Environment of an RCCL process group that may see hipGraph captures (call BEFORE init_process_group: the process group
    reads both variables when it is constructed).
    * TORCH_NCCL_TRACE_BUFFER_SIZE: the flight recorder, whose per-collective 'retired' flag is what drain_watchdog() polls;
    * TORCH_NCCL_RETHROW_CUDA_ERRORS=0, ONLY for in-graph collectives: the watchdog thread polls the completion events of eager
      collectives; HIP refuses the query of an event whose stream has meanwhile gone into capture and the watchdog would turn
      that into an abort.  drain_watchdog() empties the watchdog's list before every capture, so the situation should not arise;
      the variable is the second line of defence.  Its price, stated plainly: in this mode a genuine asynchronous device error
      seen by the watchdog is LOGGED, not thrown -- the run continues until the next synchronising call reports it."""
    if ingraph is None:
        ingraph = os.environ.get('GENRL_DP_INGRAPH', '1') != '0'
    os.environ.setdefault('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000')
    if ingraph:
        os.environ.setdefault('TORCH_NCCL_RETHROW_CUDA_ERRORS', '0')


_captured_ids = set()     # flight-recorder entries of collectives issued DURING a capture: the watchdog never tracks (or retires) those


def _trace_entries():
    import pickle
    tr = pickle.loads(torch._C._distributed_c10d._dump_nccl_trace())
    ents = tr.get('entries', []) if isinstance(tr, dict) else tr
    return [(e.get('record_id', (e.get('pg_id'), e.get('collective_seq_id'), e.get('p2p_seq_id'), e.get('op_id'))), bool(e.get('retired', True)))
            for e in ents]


def note_captured():
    """Called when a capture ends (graph.GraphedStep._end): every collective that is not retired NOW was issued inside the capture
    (drain_watchdog() emptied the list before the capture began) and will never be retired -- drain_watchdog() ignores it from here on."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'):
        return
    try:
        _captured_ids.update(i for i, retired in _trace_entries() if not retired)
    except Exception:
        pass


def drain_watchdog(timeout=2.0):
    """Block until the RCCL process group's watchdog holds no eager collective any more: device synchronise, then poll the
    flight recorder until every recorded eager collective is marked retired (the watchdog sets the flag when it drops the work from
    its list).  Deterministic where the recorder is available (TORCH_NCCL_TRACE_BUFFER_SIZE > 0, prepare_nccl_env); otherwise, and
    on a timeout, synchronise + two watchdog polling periods as before.  -> True when the list was seen empty."""
    import time
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'):
        return True
    torch.cuda.synchronize()
    try:
        if int(os.environ.get('TORCH_NCCL_TRACE_BUFFER_SIZE', '0')) > 0:
            t0 = time.time()
            while time.time() - t0 < timeout:
                if all(retired or i in _captured_ids for i, retired in _trace_entries()):
                    return True
                time.sleep(0.01)
    except Exception:
        pass
    time.sleep(0.25)
    return False


def shard_batch(batch, rank, world):
    """Rank r takes sequences [r*B/P, (r+1)*B/P) of the replay batch."""
    if world == 1:
        return batch
    out = {}
    for k, v in batch.items():
        B = v.shape[0]
        assert B % world == 0, f'batch {B} not divisible by world {world}'
        per = B // world
        out[k] = v[rank * per:(rank + 1) * per].contiguous()
    return out


def shard_rows(x, rank, world, dim):
    """Slice of a noise tensor whose `dim` enumerates global rows b-major (b*rows_per_b + i)."""
    if world == 1:
        return x
    n = x.shape[dim]
    per = n // world
    return x.narrow(dim, rank * per, per).contiguous()


_inflight = []      # asynchronous collectives not yet waited for


def _serialise():
    """gloo runs asynchronous collectives on worker threads with no ordering between them: two outstanding ones can be
    matched crosswise between ranks (hang / wrong data).  RCCL orders collectives on its stream, nothing to do there."""
    if dist.get_backend() != 'nccl':
        while _inflight:
            _inflight.pop(0).wait()


def grad_reduce(flat_grad):
    """Sum-all-reduce one flat gradient buffer in place; returns the divisor (world size).
    Under hipGraph capture the collective is a cut between two graphs (genrl_amd/graph.py)."""
    from . import graph

    def run():
        _serialise()
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    graph.cut(run)
    return dist.get_world_size()


def grad_reduce_async(flat_grad):
    """Start the sum-all-reduce of one flat gradient buffer; -> (wait, world).  The collective runs on the backend's own
    stream (RCCL: ordered after the work already enqueued on the current stream) beside whatever the caller enqueues next;
    wait() orders the current stream behind it.  Under hipGraph capture both ends are cuts between graphs."""
    from . import graph
    box = {}

    def start():
        _serialise()
        box['w'] = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=True)
        _inflight.append(box['w'])

    def wait():
        w = box.pop('w')
        if w in _inflight:
            _inflight.remove(w)
        w.wait()
    graph.cut(start)
    return (lambda: graph.cut(wait)), dist.get_world_size()


def all_gather_flat(x):
    """Concatenation of `x` over ranks.  The output buffer is allocated once per call site and is
    static across graph replays."""
    from . import graph
    x = x.contiguous()
    out = torch.empty(dist.get_world_size() * x.numel(), dtype=x.dtype, device=x.device)
    def run():
        _serialise()
        dist.all_gather_into_tensor(out, x)
    graph.cut(run)
    return out


def install(optimizer_cls, reward_ema_cls, force=False):
    """Hook the collectives into the product classes (no-op for world size 1 unless force=True: the 1-rank RCCL test
    drives the real hooks, captured collectives included, on a single GPU)."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        optimizer_cls.grad_reduce = staticmethod(grad_reduce)
        if os.environ.get('GENRL_DP_ASYNC', '1') != '0':
            optimizer_cls.grad_reduce_async = staticmethod(grad_reduce_async)
        reward_ema_cls.all_gather = staticmethod(all_gather_flat)
        # RCCL orders every collective of the process group on its own stream: the connector's side stream
        # (cfg.overlap_detached) may issue its reductions beside the main stream's (gloo's worker threads give no such order)
        optimizer_cls.overlap_under_dp = dist.get_backend() == 'nccl' and os.environ.get('GENRL_DP_OVERLAP', '1') != '0'
        # The fused Dense -> LayerNorm launches (planes.gemm_ln) wait for their peer workgroups to be RESIDENT.  Beside RCCL's persistent
        # kernels (reductions run beside the next phase) -- or beside a second rank sharing the GPU, as in the gloo rehearsals -- that
        # residency has never been exercised on hardware here: with more than one rank the layers keep their two launches (-0.1 ms of
        # a 10-15 ms per-rank step given up; GENRL_GEMM_LN=2 keeps the fused form).
        if dist.get_world_size() > 1 and os.environ.get('GENRL_GEMM_LN') != '2':
            from . import planes
            global _ln_fused_before
            _ln_fused_before = planes.LN_FUSED if _ln_fused_before is None else _ln_fused_before
            planes.LN_FUSED = False


_ln_fused_before = None


def uninstall(optimizer_cls, reward_ema_cls):
    global _ln_fused_before
    if _ln_fused_before is not None:
        from . import planes
        planes.LN_FUSED, _ln_fused_before = _ln_fused_before, None
    optimizer_cls.grad_reduce = None
    optimizer_cls.grad_reduce_async = None
    optimizer_cls.overlap_under_dp = False
    reward_ema_cls.all_gather = None


def barrier_max(seconds, device):
    """max over ranks of a host-measured duration"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
