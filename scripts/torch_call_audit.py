"""Which agent-code lines issue plain torch ops in one training step (each is at least one ~5 us launch):
TorchFunctionMode around one eager step, calls attributed to the innermost genrl_amd/ frame.  [topN]"""
import sys, os, collections, contextlib, traceback
sys.path.insert(0, '.')
import torch
from torch.overrides import TorchFunctionMode
import bench
from genrl_amd import config

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
SKIP = {'__get__', 'size', 'dim', 'shape', 'stride', 'is_contiguous', 'data_ptr', 'numel', 'requires_grad_', 'detach', 'view',
        'reshape', 'unsqueeze', 'squeeze', 'permute', 'transpose', 'expand', '__getitem__', 'is_floating_point', 'apply',
        'contiguous', 'flatten', 'unflatten', 'chunk', 'split', 'type', 'to', '__set__', 'is_cuda', 'element_size'}
by = collections.Counter()
class Audit(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, '__name__', str(func))
        if name not in SKIP:
            fr = None
            for f in reversed(traceback.extract_stack(limit=12)):
                if 'genrl_amd' in f.filename and 'torch' not in f.filename.split('genrl_amd')[0][-8:]:
                    fr = f; break
            where = f'{os.path.relpath(fr.filename)}:{fr.lineno}' if fr else '?'
            by[(where, name)] += 1
        return func(*args, **(kwargs or {}))
with Audit():
    bench.one_step(ag, batch)
torch.cuda.synchronize()
lines = collections.Counter()
for (w, n), c in by.items():
    lines[w] += c
print('torch-level calls in one step (forward side):', sum(by.values()))
for w, c in lines.most_common(top):
    ops = ', '.join(f'{n}x{k}' for (ww, n), k in sorted(by.items(), key=lambda kv: -kv[1]) if ww == w)
    print(f'{c:4d}  {w:44s} {ops[:110]}')
