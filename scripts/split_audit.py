"""who calls planes.split / ops_conv_planes._uniform_split in ONE eager training step (shapes, innermost genrl_amd line): GPU box only"""
import sys, os, collections, traceback, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config, planes, ops_conv_planes

cfgname = sys.argv[1] if len(sys.argv) > 1 else 'c2'
dev = 'cuda:0'
cfg = config.default_cfg(32, 32, device=dev, overlap_detached=False)
with contextlib.redirect_stdout(sys.stderr):
    ag = config.make_agent(cfg)
ag.wm.viclip_model = bench.TextStub()
batch = {k: torch.from_numpy(v).to(dev) for k, v in bench.synth_batch(32, 32).items()}
for _ in range(2):
    bench.one_step(ag, batch)
torch.cuda.synchronize()
by = collections.Counter()


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(x2d, *a, **kw):
        fr = [f_ for f_ in traceback.extract_stack(limit=12)[:-1] if '/genrl_amd/' in f_.filename]
        where = ' <- '.join(f'{os.path.basename(f_.filename)}:{f_.lineno}' for f_ in reversed(fr[-3:]))
        by[(name, tuple(x2d.shape), bool(kw.get('transpose', a[0] if a else False)), where)] += 1
        return orig(x2d, *a, **kw)
    setattr(mod, name, f)


wrap(planes, 'split')
wrap(ops_conv_planes, '_uniform_split')
bench.one_step(ag, batch)
torch.cuda.synchronize()
print('split calls in one step:', sum(by.values()))
for (n, shp, tr, w), c in sorted(by.items(), key=lambda kv: -kv[1]):
    print(f'{c:3d}  {n:15s} {str(shp):18s} transpose={int(tr)}  {w}')
