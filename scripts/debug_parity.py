import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import test_gpu_iteration as T
from genrl_amd import config
from oracle.iteration import run_iteration
name = sys.argv[1] if len(sys.argv) > 1 else 'tiny_iter.npz'
tiny = name.startswith('tiny')
tiny_o = dict(deter=32, hidden=32, units=32, cnn_depth=4) if tiny else {}
g, ocfg, p, batch, noise, ag, outputs, mets_wm, mets, grads = T.run_product(name, True, config.tiny_overrides() if tiny else {}, tiny_o)
for key, val in g.items():
    for pre, src in (('metrics_wm.', mets_wm), ('metrics_conn2.', mets), ('metrics_imag.', mets)):
        if key.startswith(pre):
            n = key[len(pre):]
            if pre == 'metrics_wm.' and ('connector' in n or 'aligner' in n): continue
            rel = abs(src[n] - float(val)) / (abs(float(val)) + 1e-12)
            print(f'{n:35s} {src[n]: .7e} ref {float(val): .7e} rel {rel:.2e}')
if tiny:
    res = run_iteration(p, ocfg, batch, noise, T.FakeClip().get_txt_feat(''), apply_updates=False)
    for ph in ('wm', 'conn1', 'actor', 'critic'):
        worst = []
        for nme, gref in res['grads'][ph].items():
            a, b = grads[ph][nme].double(), gref.double()
            worst.append(((a - b).norm().item() / (b.norm().item() + 1e-30), nme))
        worst.sort(reverse=True)
        print(ph, [(f'{w:.1e}', n) for w, n in worst[:4]])
    print('imag idx mism', None)
