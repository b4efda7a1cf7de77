"""precision 16: per-parameter gradient differences product vs the oracle's bf16-operand mode (localises a product that rounds differently)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from test_gpu_iteration import run_product, FakeClip, _restore_fp32_arithmetic
from oracle import genrl_oracle as O
from oracle.iteration import run_iteration
torch.set_num_threads(16)
meta = {'meta': (4, 16, 10, 32, 32, 16, 5), 'img': 64}
try:
    g, ocfg, p, batch, noise, ag, outputs, w16, m16, grads = run_product(meta, True, dict(precision=16), {})
finally:
    _restore_fp32_arithmetic()
with O.bf16_operands():
    res = run_iteration(p, ocfg, batch, noise, FakeClip().get_txt_feat(''), apply_updates=False)
om = {k: float(v) for k, v in res['metrics'].items()}
for k, v in {**w16, **m16}.items():
    if k in om and np.isfinite(om[k]) and abs(v - om[k]) > 2e-4 * abs(om[k]) + 1e-6:
        print('metric', k, v, om[k])
for ph in ('wm', 'conn1', 'conn2', 'actor', 'critic'):
    for name, gref in res['grads'][ph].items():
        a, b = grads[ph][name].double(), gref.double()
        rel = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        if rel > 1e-3:
            print(f'{ph:6s} {name:60s} rel L2 err {rel:.2e}  |ref| {b.norm().item():.3e}')
print('done')
