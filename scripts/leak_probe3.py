"""who keeps the autograd graphs of past eager iterations alive: tensors that still carry a grad_fn after the iterations are over"""
import sys, os, contextlib, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config
cfg = config.default_cfg(32, 32, device='cuda:0', overlap_detached=False)
with contextlib.redirect_stdout(sys.stderr):
    ag = config.make_agent(cfg)
ag.wm.viclip_model = bench.TextStub()
batch = {k: torch.from_numpy(v).to('cuda:0') for k, v in bench.synth_batch(32, 32).items()}
for i in range(3):
    m = bench.one_step(ag, batch)
del m
torch.cuda.synchronize(); gc.collect()
live = [o for o in gc.get_objects() if torch.is_tensor(o) and o.grad_fn is not None]
print(len(live), 'tensors with a grad_fn alive')
for o in live[:40]:
    refs = [r for r in gc.get_referrers(o) if r is not live and not isinstance(r, type(sys._getframe()))]
    desc = []
    for r in refs[:4]:
        if isinstance(r, dict):
            owners = [type(q).__name__ for q in gc.get_referrers(r) if not isinstance(q, (dict, list, tuple))][:3]
            desc.append(f'dict{list(r)[:6]} owned by {owners}')
        else:
            desc.append(type(r).__name__)
    print(tuple(o.shape), type(o.grad_fn).__name__, '<-', desc)
