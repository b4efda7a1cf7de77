"""eager iterations: device memory allocated after every 10 steps (a flat line is the expectation).  python scripts/leak_probe.py [steps] [c2|c3|c4|c5]"""
import sys, os, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 41
wl = sys.argv[2] if len(sys.argv) > 2 else 'c2'
with contextlib.redirect_stdout(sys.stderr):
    if wl == 'c3':
        B, T, A, img = 16, 50, 6, 64
        ag = config.make_dreamer_agent(config.dreamer_cfg(B, T, device='cuda:0'), act_dim=A); step = bench.dreamer_step
    elif wl == 'c4':
        B, T, A, img = 8, 32, 9, 128
        ag = config.make_agent(config.default_cfg(B, T, device='cuda:0', encoder=dict(cnn_kernels=[4] * 5), decoder=dict(cnn_kernels=[5, 5, 5, 6, 6])),
                               act_dim=A, img=img); step = bench.one_step
    elif wl == 'c5':
        B, T, A, img = 16, 16, 10, 64
        ag = config.make_agent(config.default_cfg(B, T, device='cuda:0', imag_horizon=15), act_dim=A); step = bench.datafree_step
    else:
        B, T, A, img = 32, 32, 10, 64
        ag = config.make_agent(config.default_cfg(B, T, device='cuda:0', overlap_detached=True)); step = bench.one_step
if wl != 'c3':
    ag.wm.viclip_model = bench.TextStub()
full = bench.synth_batch(B, T, A=A, img=img)
if wl == 'c3':
    full.pop('clip_video')
batch = {} if wl == 'c5' else {k: torch.from_numpy(v).to('cuda:0') for k, v in full.items()}
out = []
for i in range(steps):
    step(ag, batch)
    if i % 10 == 0:
        torch.cuda.synchronize()
        out.append(round(torch.cuda.memory_allocated() / 2 ** 30, 2))
print(os.environ.get('TAG', wl), 'GiB allocated every 10 steps:', out, flush=True)
