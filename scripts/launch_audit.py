"""Which Python lines launch the step's small kernels: one eager step under torch.profiler (with_stack), every
kernel-launching CPU op attributed to its innermost genrl_amd/ frame.  scripts/launch_audit.py [topN]"""
import sys, os, collections, contextlib
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from genrl_amd import config
from torch.profiler import profile, ProfilerActivity

top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = 'cuda:0'
cfg = config.default_cfg(32, 32, device=dev, overlap_detached=False)
with contextlib.redirect_stdout(sys.stderr):
    ag = config.make_agent(cfg)
ag.wm.viclip_model = bench.TextStub()
batch = {k: torch.from_numpy(v).to(dev) for k, v in bench.synth_batch(32, 32).items()}
for _ in range(2):
    bench.one_step(ag, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench.one_step(ag, batch)
    torch.cuda.synchronize()
# kernel launches are recorded as runtime events (hipLaunchKernel / hipExtModuleLaunchKernel) nested in CPU ops
evs = prof.events()
by = collections.Counter()
def chain(e):
    out = []
    p = e.cpu_parent
    while p is not None and len(out) < 3:
        if not p.name.startswith('aten::') and 'ProfilerStep' not in p.name:
            out.append(p.name[:48])
        p = p.cpu_parent
    return ' < '.join(out) if out else '(top level)'
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if e.cpu_children and any(c.kernels for c in e.cpu_children):
        continue                                     # attribute to the innermost op only
    frame = next((f for f in (e.stack or []) if 'genrl_amd' in f or 'bench.py' in f), '')
    by[(e.name[:40], chain(e), frame.replace(os.getcwd() + '/', '')[:60])] += len(e.kernels)
print('kernel launches in one eager step:', sum(by.values()))
for (n, c, f), k in by.most_common(top):
    print(f'{k:5d}  {n:40s} in {c}  {f}')
