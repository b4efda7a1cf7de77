"""Every aten operator that reaches the device in ONE eager training step (forward, autograd engine and optimiser side), with its
argument shapes and -- for calls made from Python -- the innermost genrl_amd/ line: the torch-native launches left in the step.
GPU box only: python scripts/aten_audit.py [topN]"""
import sys, os, collections, contextlib, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from genrl_amd import config

top = int(sys.argv[1]) if len(sys.argv) > 1 else 80
dev = 'cuda:0'
cfg = config.default_cfg(32, 32, device=dev, overlap_detached=False)
with contextlib.redirect_stdout(sys.stderr):
    ag = config.make_agent(cfg)
ag.wm.viclip_model = bench.TextStub()
batch = {k: torch.from_numpy(v).to(dev) for k, v in bench.synth_batch(32, 32).items()}
for _ in range(2):
    bench.one_step(ag, batch)
torch.cuda.synchronize()
VIEW = ('view', 'reshape', 'alias', 'detach', 'expand', 'permute', 'transpose', 'select', 'slice', 'squeeze', 'unsqueeze', 'as_strided', 't.',
        'unbind', 'split', '_unsafe_view', 'empty', 'unfold', 'sym_', 'size', 'stride', 'numel', 'is_', 'dim', 'lift_fresh', 'chunk', 'narrow',
        'new_empty', 'storage_offset', 'unflatten', 'flatten', 'set_', 'resize_', '_local_scalar_dense', 'item', 'record_stream', 'contiguous')
by = collections.Counter()


def shp(a):
    if isinstance(a, torch.Tensor):
        return 'x'.join(map(str, a.shape)) or 's'
    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
        return '[' + ','.join(shp(t) for t in a[:3]) + (',..' if len(a) > 3 else '') + ']'
    return None


class Audit(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if not any(name.startswith(v) for v in VIEW):
            fr = None
            for f in reversed(traceback.extract_stack(limit=16)):
                if '/genrl_amd/' in f.filename:
                    fr = f; break
            where = f'{os.path.relpath(fr.filename)}:{fr.lineno}' if fr else '(autograd engine)'
            by[(where, name, ' '.join(s for s in map(shp, args) if s))] += 1
        return func(*args, **(kwargs or {}))


with Audit():
    bench.one_step(ag, batch)
torch.cuda.synchronize()
print('non-view aten calls in one step:', sum(by.values()))
for (w, n, s), c in sorted(by.items(), key=lambda kv: (-kv[1], kv[0]))[:top]:
    print(f'{c:4d}  {w:46s} {n:34s} {s[:80]}')
