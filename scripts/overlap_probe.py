"""How much of the connector updates is hidden behind the main chain?  Graph-replayed step time of the bench workload in four variants
on ONE box: full (side stream on / off), without the connector updates, without the imagination phase.  Diagnostic only (the variants
without a phase are not the workload).  GPU box only: python scripts/overlap_probe.py [batch=32]"""
import sys, os, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config
from genrl_amd.graph import GraphedStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def run(overlap, conn=True, imag=True, steps=30):
    torch.manual_seed(0)
    cfg = config.default_cfg(B, 32, device='cuda', overlap_detached=overlap)
    with contextlib.redirect_stdout(sys.stderr):
        ag = config.make_agent(cfg)
    ag.wm.viclip_model = bench.TextStub()
    batch = {k: torch.from_numpy(v).cuda() for k, v in bench.synth_batch(B, 32).items()}
    if not conn:
        ag.wm.update_additional_detached_modules = lambda data, outputs, metrics: (0, metrics)

    def step(ag_, batch_):
        state, outputs, mets = ag_.update_wm(batch_, 0)
        _, mets = ag_.wm.update_additional_detached_modules(batch_, outputs, mets)
        if imag:
            _, mets = ag_.update_imag_behavior(state=None, outputs=outputs, metrics=mets, seq_data=batch_)
        else:
            from genrl_amd import streams
            streams.join()
        return mets
    gs = GraphedStep(ag, batch, step, warmup=2)
    for _ in range(5): gs()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): gs()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


full_on, full_off = run(True), run(False)
no_conn = run(True, conn=False)
no_imag_off = run(False, imag=False)
no_imag_no_conn = run(False, conn=False, imag=False)
print(f'B={B}: full, side stream on {full_on:.2f} ms | off {full_off:.2f} ms | without the connector updates {no_conn:.2f} ms | '
      f'world model + connector only {no_imag_off:.2f} ms | world model only {no_imag_no_conn:.2f} ms')
print(f'  connector updates alone (serial): {full_off - no_conn:.2f} ms; still exposed with the side stream: {full_on - no_conn:.2f} ms; '
      f'imagination phase: {no_conn - no_imag_no_conn:.2f} ms')
