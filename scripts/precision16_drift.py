import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import test_gpu_iteration as T
from genrl_amd import ops
meta = {'meta': (32, 32, 10, 32, 32, 16, 3)}
_, _, _, _, _, _, _, w32, m32, g32 = T.run_product(meta, True, {}, {})
_, _, _, _, _, _, _, w16, m16, g16 = T.run_product(meta, True, dict(precision=16), {})
ops.set_gemm_precision('f32')
worst = 0
for k in sorted(m32):
    if k in m16 and np.isfinite(m32[k]) and abs(m32[k]) > 1e-8:
        r = abs(m16[k] - m32[k]) / abs(m32[k]); worst = max(worst, r)
        print(f'{k:36s} f32 {m32[k]: .6e} bf16 {m16[k]: .6e} rel {r:.2e}')
for ph in g32:
    a = np.sqrt(sum(float((t.double() ** 2).sum()) for t in g16[ph].values()))
    b = np.sqrt(sum(float((t.double() ** 2).sum()) for t in g32[ph].values()))
    d = np.sqrt(sum(float(((g16[ph][n] - g32[ph][n]).double() ** 2).sum()) for n in g32[ph]))
    print(f'grad {ph}: norm f32 {b:.5e} bf16 {a:.5e} rel-diff-of-norms {abs(a-b)/b:.2e} |g16-g32|/|g32| {d/b:.2e}')
print('worst metric rel', worst)
