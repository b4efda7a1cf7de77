"""hipGraph capture of one whole training iteration.

The hot path launches ~1.5 k kernels per iteration, many of them a few microseconds long (the
sequential RSSM / imagination chains).  Eager Python cannot always feed those fast enough — above
all when the replay batch is sharded over 8 GPUs and each rank's kernels shrink while the host work
per iteration stays the same.  `GraphedStep` captures the iteration once into hipGraphs
(`torch.cuda.CUDAGraph`: our ctypes launches go to torch's current stream, which is the capturing
stream) and replays them: a handful of host calls per iteration, kernels back-to-back.

What makes the iteration capturable: no host synchronisation anywhere on the path (metrics stay
device tensors), sampling noise from the device generator, Adam's step count on the device
(`FlatGroup.step_dev`), workspaces from torch's graph-private pool.

Collectives, two modes (`GraphedStep(collectives=...)`):
* 'ingraph' (RCCL only): the all-reduces / all-gathers that genrl_amd/dp.py issues are captured INSIDE the one
  graph -- torch's NCCL process group records them on its own stream, which the capture follows through the
  event edges, so an asynchronous reduction becomes a graph branch beside the next phase's kernels and the
  connector's side stream (cfg.overlap_detached) stays usable under data parallelism.  `cut()` runs inline.
* 'cut' (any backend; the fallback): the capture is split into consecutive graphs sharing one memory pool and
  the collective runs eagerly between two replays on buffers that are static across replays (flat gradient
  buffers; a preallocated gather output).  `cut()` is called by genrl_amd/dp.py's hooks.

The only host-side decision of the reference's iteration — the slow-critic hard copy every
`slow_target_update` updates (agent/dreamer.py:455-462) — is deferred to after the replay."""
import torch

_active = None      # the GraphedStep currently capturing (None otherwise)
_abandoned = []     # graphs whose capture failed: never destroyed (destroying a graph whose stream is still capturing aborts the process)


def cut(eager_fn):
    """Run `eager_fn` outside graph capture.  Inside a capture: close the current graph, run the
    function eagerly (and remember it for every replay), open the next graph."""
    if _active is None or _active.collectives == 'ingraph':
        return eager_fn()
    return _active._cut(eager_fn)


def cutting():
    """True while a capture is running whose collectives are cuts: a cut ends the capture on the capturing stream, so nothing
    that contains a collective may sit on a forked side stream then (WorldModel.update_additional_detached_modules)"""
    return _active is not None and _active.collectives == 'cut'


class GraphedStep:
    def __init__(self, agent, batch, step_fn, warmup=3, collectives='cut'):
        """batch: dict of device tensors whose storage becomes the graphs' static input.
        collectives: 'cut' (eager between graph segments) or 'ingraph' (captured; RCCL only)."""
        global _active
        assert collectives in ('cut', 'ingraph')
        self.collectives = collectives
        self.agent = agent
        self.static_batch = {k: v.clone() for k, v in batch.items()}
        self.step_fn = step_fn
        self.items = []          # ('graph', CUDAGraph) | ('eager', fn)
        self.metrics = None
        ac = self._behavior(agent)
        ac._defer_slow_target = True
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(warmup):
                step_fn(agent, self.static_batch)
                ac.update_slow_target()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.pool = torch.cuda.graph_pool_handle()
        self._g = None
        steps_before = [g.step for g in self._groups()]
        _active = self
        try:
            with torch.cuda.stream(self.stream):
                # every cached weight-plane set counts as stale when the capture starts: the captured iteration then re-splits
                # each one at its first use (all of them change once per iteration anyway), whatever the warm-up's history of
                # creations and refreshes left marked fresh -- a replay must never depend on that history
                from . import planes
                planes.invalidate()
                self._begin()
                self.metrics = step_fn(agent, self.static_batch)
                self._end()
        except BaseException:
            # leave no stream in capture mode and no half-built state behind: the caller may fall back to eager steps
            if self._g is not None:
                _abandoned.append(self._g)
                try:
                    self._g.capture_end()
                except Exception:
                    pass
                self._g = None
            _abandoned.extend(it for kind, it in self.items if kind == 'graph')
            self.items = []
            ac._defer_slow_target = False
            raise
        finally:
            _active = None
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        # the captured iteration did not execute: undo the host-side bookkeeping its Python ran (optimiser step
        # counters -- per group by what the capture really added: the connector's group steps twice per iteration, SURVEY
        # Q1 -- ; the slow-critic cadence is advanced by __call__ only, once per replay)
        self._groups_captured = self._groups()
        self._step_delta = [g.step - b for g, b in zip(self._groups_captured, steps_before)]
        for g, d in zip(self._groups_captured, self._step_delta):
            g.step -= d

    @staticmethod
    def _drain_backend():
        """RCCL: the process group's watchdog thread polls the completion event of every EAGER collective still on its list
        (hipEventQuery, every 100 ms); HIP refuses that query -- 'operation not permitted on an event last recorded in a capturing
        stream', which the watchdog turns into an abort of the process -- once the stream the event was recorded on is capturing.
        Collectives issued during a capture are not put on that list, so it is enough to let the list run empty before a capture
        begins: dp.drain_watchdog() synchronises the device and polls the flight recorder until every collective is retired."""
        from . import dp
        dp.drain_watchdog()

    def _begin(self):
        self._drain_backend()
        self._g = torch.cuda.CUDAGraph()
        # thread_local: collectives started at a cut may still be progressing on the backend's own threads (gloo's workers,
        # the RCCL watchdog) while the next graph is being captured; only THIS thread's calls belong to the capture
        self._g.capture_begin(pool=self.pool, capture_error_mode='thread_local')

    def _end(self):
        self._g.capture_end()
        from . import dp
        dp.note_captured()
        self.items.append(('graph', self._g))
        self._g = None

    def _cut(self, eager_fn):
        self._end()
        out = eager_fn()
        self.items.append(('eager', eager_fn))
        self._begin()
        return out

    def __call__(self, batch=None):
        if batch is not None:
            for k, v in batch.items():
                self.static_batch[k].copy_(v, non_blocking=True)
        for kind, it in self.items:
            if kind == 'graph':
                it.replay()
            else:
                it()
        for g, d in zip(self._groups_captured, self._step_delta):
            g.step += d
        self._behavior(self.agent).update_slow_target()
        return self.metrics

    @staticmethod
    def _behavior(agent):
        """the actor-critic the iteration trains: GenRLAgent's imagination behaviour, DreamerAgent's acting behaviour"""
        b = getattr(agent, '_imag_behavior', None)
        return b if b is not None else agent._acting_behavior

    def _groups(self):
        """every flat optimiser group the iteration steps (the connector's lives in wm.model_opt beside the world model's)"""
        ag = self.agent
        ac = self._behavior(ag)
        opts = [ag.wm.model_opt, ac.actor_opt, ac.critic_opt]
        return [g for o in opts for g in o._groups]
