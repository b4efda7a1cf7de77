"""Which parameters still get their gradient through autograd's AccumulateGrad node (a tensor returned by a backward) instead of
the direct flat-buffer epilogues?  python scripts/accgrad_audit.py [c2|c3]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from genrl_amd import config
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
B, T = (4, 16)
if wl == 'c3':
    cfg = config.dreamer_cfg(B, T, device='cuda:0')
    ag = config.make_dreamer_agent(cfg, act_dim=6)
    step, A = bench.dreamer_step, 6
else:
    cfg = config.default_cfg(B, T, device='cuda:0', overlap_detached=True)
    ag = config.make_agent(cfg)
    ag.wm.viclip_model = bench.TextStub()
    step, A = bench.one_step, 10
full = bench.synth_batch(B, T, A=A)
if wl == 'c3':
    full.pop('clip_video')
batch = {k: torch.from_numpy(v).cuda() for k, v in full.items()}
hits = collections.Counter()
names = {}
keep = []
for n, p in ag.named_parameters():
    names[id(p)] = n
    rg = p.requires_grad
    p.requires_grad_(True)
    node = p.view_as(p).grad_fn.next_functions[0][0]          # the parameter's AccumulateGrad node
    keep.append(node)
    def pre(grads, n=n):
        if grads[0] is not None:
            hits[n] += 1
    node.register_prehook(pre)
    p.requires_grad_(rg)
step(ag, batch); torch.cuda.synchronize()
hits.clear()
step(ag, batch); torch.cuda.synchronize()
print(f'{len(hits)} parameters of {len(names)} go through AccumulateGrad in one {wl} iteration:')
for n, c in sorted(hits.items()):
    print(f'  {c}x {n}')
