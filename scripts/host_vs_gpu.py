import sys, time, os
sys.path.insert(0, '.')
import torch
from bench import synth_batch, TextStub, one_step
from genrl_amd import config
cfg = config.default_cfg(32, 32, device='cuda')
torch.manual_seed(0)
ag = config.make_agent(cfg); ag.wm.viclip_model = TextStub()
batch = {k: torch.from_numpy(v).cuda() for k, v in synth_batch(32, 32).items()}
for _ in range(3): one_step(ag, batch)
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter(); one_step(ag, batch); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms')
# phase split (synchronised)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); t1 = time.perf_counter(); torch.cuda.synchronize(); return r, 1e3*(t1-t0), 1e3*(time.perf_counter()-t0)
(st, host, tot) = timed(lambda: ag.update_wm(batch, 0)); state, outputs, mets = st
print(f'update_wm (wm + conn1): host {host:.1f} total {tot:.1f}')
(_, host, tot) = timed(lambda: ag.wm.update_additional_detached_modules(batch, outputs, mets))
print(f'connector 2: host {host:.1f} total {tot:.1f}')
(_, host, tot) = timed(lambda: ag.update_imag_behavior(state=None, outputs=outputs, metrics=mets, seq_data=batch))
print(f'imag behaviour: host {host:.1f} total {tot:.1f}')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); one_step(ag, batch); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
