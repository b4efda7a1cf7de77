"""Which call sites still launch the fp32-operand GEMM (ops.sgemm) in one eager training step, by shape: the products that are
not on plane operands yet.  GPU box only: python scripts/sgemm_callers.py [min_M]"""
import sys, os, collections, contextlib, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config, ops

minM = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
orig = ops.sgemm


def spy(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, *a, **k):
    if M >= minM:
        fr = [f for f in traceback.extract_stack(limit=14) if '/genrl_amd/' in f.filename][-4:]
        by[(M, N, K, ('k' if a_ks == 1 else 'r') + ('k' if b_ks == 1 else 'r'),
            ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(fr)))] += 1
    return orig(A, a_rs, a_ks, B, b_rs, b_ks, C, ldc, bias, M, N, K, *a, **k)


import genrl_amd
for name, mod in list(sys.modules.items()):
    if name.startswith('genrl_amd') and mod is not None and getattr(mod, 'sgemm', None) is orig:
        mod.sgemm = spy
bench.one_step(ag, batch)
torch.cuda.synchronize()
for (M, N, K, t, w), c in sorted(by.items(), key=lambda kv: -kv[1] * kv[0][0] * kv[0][1] * kv[0][2]):
    print(f'{c:3d} x {M:7d} {N:6d} {K:7d} {t}  {w}')
