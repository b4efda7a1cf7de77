"""eager iterations: device memory allocated after every 10 steps (a flat line is the expectation)"""
import sys, os, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config
cfg = config.default_cfg(32, 32, device='cuda:0', overlap_detached=True)
with contextlib.redirect_stdout(sys.stderr):
    ag = config.make_agent(cfg)
ag.wm.viclip_model = bench.TextStub()
batch = {k: torch.from_numpy(v).to('cuda:0') for k, v in bench.synth_batch(32, 32).items()}
out = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 41):
    bench.one_step(ag, batch)
    if i % 10 == 0:
        torch.cuda.synchronize()
        out.append(round(torch.cuda.memory_allocated() / 2 ** 30, 2))
print(os.environ.get('TAG', ''), 'GiB allocated every 10 steps:', out, flush=True)
