"""which phase of the eager iteration retains memory: GiB allocated after every 5 calls of a truncated iteration"""
import sys, os, contextlib, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config
mode = sys.argv[1]
cfg = config.default_cfg(32, 32, device='cuda:0', overlap_detached=(mode != 'full-nooverlap'))
with contextlib.redirect_stdout(sys.stderr):
    ag = config.make_agent(cfg)
ag.wm.viclip_model = bench.TextStub()
batch = {k: torch.from_numpy(v).to('cuda:0') for k, v in bench.synth_batch(32, 32).items()}
out = []
for i in range(16):
    state, outputs, mets = ag.update_wm(batch, 0)
    if mode != 'wm':
        _, mets = ag.wm.update_additional_detached_modules(batch, outputs, mets)
        if mode == 'wm+conn':
            from genrl_amd import streams; streams.join()
    if mode.startswith('full'):
        _, mets = ag.update_imag_behavior(state=None, outputs=outputs, metrics=mets, seq_data=batch)
    del state, outputs, mets
    if i % 5 == 0:
        torch.cuda.synchronize(); gc.collect()
        out.append(round(torch.cuda.memory_allocated() / 2 ** 30, 2))
print(mode, out, flush=True)
if mode == 'full':
    # what is alive: the largest tensors by storage
    import collections
    sizes = collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                sizes[(tuple(o.shape), o.dtype)] += 1
        except Exception:
            pass
    top = sorted(sizes.items(), key=lambda kv: -kv[1] * torch.empty(kv[0][0], dtype=kv[0][1]).numel() if len(kv[0][0]) else 0)[:12]
    for (shape, dt), c in top:
        print('   ', c, 'x', shape, dt)
