"""hipGraph capture of one whole training iteration.

The hot path launches ~5 k kernels per iteration, many of them a few microseconds long (the
sequential RSSM / imagination chains).  Eager Python cannot feed those fast enough, so the GPU
idles between launches.  `GraphedStep` captures the iteration once into a hipGraph
(`torch.cuda.CUDAGraph`: our ctypes launches go to torch's current stream, which is the capturing
stream) and replays it: one host call per iteration, kernels back-to-back.

What makes the iteration capturable: no host synchronisation anywhere on the path (metrics stay
device tensors), sampling noise from the device generator, Adam's step count on the device
(`FlatGroup.step_dev`), workspaces from torch's graph-private pool.  The only host-side decision
of the reference's iteration — the slow-critic hard copy every `slow_target_update` updates
(agent/dreamer.py:455-462) — is deferred to `after_replay`."""
import torch


class GraphedStep:
    def __init__(self, agent, batch, step_fn, warmup=3):
        """batch: dict of device tensors whose storage becomes the graph's static input."""
        self.agent = agent
        self.static_batch = {k: v.clone() for k, v in batch.items()}
        self.step_fn = step_fn
        self.graph = None
        self.metrics = None
        ac = agent._imag_behavior
        ac._defer_slow_target = True
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                step_fn(agent, self.static_batch)
                ac.update_slow_target()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.metrics = step_fn(agent, self.static_batch)
        # host-side bookkeeping the capture executed once
        ac.update_slow_target()

    def __call__(self, batch=None):
        if batch is not None:
            for k, v in batch.items():
                self.static_batch[k].copy_(v, non_blocking=True)
        self.graph.replay()
        for g in self._groups():
            g.step += 1
        self.agent._imag_behavior.update_slow_target()
        return self.metrics

    def _groups(self):
        ag = self.agent
        opts = [ag.wm.model_opt, ag._imag_behavior.actor_opt, ag._imag_behavior.critic_opt]
        return [g for o in opts for g in o._groups]
