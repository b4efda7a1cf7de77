"""The CPU oracle's full iteration at 16 / 32 / 64 / all host threads (round-5 verdict item 8: bench.py's cpu_baseline caps the leg at 16
threads -- "more only add OpenMP overhead at these sizes" -- this prints the evidence).  B8 x T32 (a quarter of configs[1]'s rows),
1 warm-up + 2 timed iterations per thread count; run on the GPU box's host:  python scripts/cpu_threads.py > profiles/r06_cpu_threads.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
import detgen
from param_shapes import agent_param_shapes
from oracle import genrl_oracle as O
from oracle.iteration import run_iteration
from bench import synth_batch, TextStub, _cpu_model

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 32)
cfg = O.make_cfg()
p = detgen.det_state_dict(agent_param_shapes(cfg), 0)
text = TextStub().get_txt_feat('')
batch = {k: torch.from_numpy(v) for k, v in synth_batch(B, T).items()}
noise = detgen.iteration_noise(B, T, cfg.stoch, cfg.discrete, cfg.act_dim, cfg.horizon)
ncpu = os.cpu_count() or 1
print(f'# CPU oracle (oracle/iteration.py, full iteration with optimiser updates) at B{B} x T{T} on {_cpu_model()} ({ncpu} logical CPUs), torch {torch.__version__}')
print('# threads  min_s  second_s  (1 warm-up + 2 timed)')
for n in [t for t in (1, 8, 16, 32, 64, 128, ncpu) if t <= ncpu]:
    if n == 1 and B * T > 256:
        continue
    torch.set_num_threads(n)
    ts = []
    for i in range(3):
        t0 = time.time()
        run_iteration(p, cfg, batch, noise, text, apply_updates=True)
        ts.append(time.time() - t0)
    ts = sorted(ts[1:])
    print(f'{n:8d}  {ts[0]:.2f}  {ts[1]:.2f}', flush=True)
