"""Do the plane-operand products (h2) change where training goes?  The same agent, the same fixed batch and the same per-site noise
(noise.static) trained for N optimiser steps (lr > 0) in three arithmetics: default (fp16-plane products) | fp32 MFMAs throughout
(planes off, GENRL_GEMM_MODE=0 semantics) | the in-register bf16 split on every tile (planes off) -- over several SEEDS (weights,
batch and noise all re-drawn per seed).  Prints the losses along the way per seed, the relative difference of every metric at the end,
and -- the yardstick -- the spread of the same metrics ACROSS seeds in the strict fp32 arithmetic.
python scripts/mode_drift.py [steps=300] [batch=32] [seeds=1]"""
import sys, os, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from genrl_amd import config, noise, ops, planes
from genrl_amd.graph import GraphedStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
keys = ('model_loss', 'kl_loss', 'observation_loss', 'reward_loss', 'connector_model_loss', 'imag_actor_loss', 'imag_critic_loss',
        'imag_reward_mean', 'imag_critic_target', 'model_grad_norm', 'imag_actor_grad_norm')
MODES = ('default', 'fp32', 'split3')


def run(seed):
    batch = {k: torch.from_numpy(v).cuda() for k, v in bench.synth_batch(B, 32, seed=1 + seed).items()}
    cache = {}
    res = {}
    for mode in MODES:
        prev_f32, prev_pl = ops.F32_MODE, planes.ENABLED          # (the agent re-selects ops.F32_MODE at every entry point)
        if mode != 'default':
            ops.F32_MODE = {'fp32': 'f32', 'split3': 'bf16x3'}[mode]
            planes.ENABLED = False
        torch.manual_seed(seed)
        cfg = config.default_cfg(B, 32, device='cuda')
        with contextlib.redirect_stdout(sys.stderr):
            ag = config.make_agent(cfg)
        ag.wm.viclip_model = bench.TextStub()
        hist = []
        with noise.static(seed=5 + seed, cache=cache):
            gs = GraphedStep(ag, batch, bench.one_step, warmup=1)
            for i in range(2, N + 1):
                m = gs()
                if i in (2, 10, 30, 100, 200, N):
                    torch.cuda.synchronize()
                    hist.append((i, {k: float(m[k]) for k in keys}))
        res[mode] = hist
        planes.ENABLED, ops.F32_MODE = prev_pl, prev_f32
        del gs, ag
    return res


rel = lambda x, y: {k: abs(x[k] - y[k]) / (abs(y[k]) + 1e-12) for k in keys[:9]}
fmt = lambda d: '{' + ', '.join(f'{k}: {v:.1e}' for k, v in d.items()) + '}'
finals = []
for seed in range(S):
    res = run(seed)
    print(f'== seed {seed}')
    for (i, a), (_, b), (_, c) in zip(res['default'], res['fp32'], res['split3']):
        print(f'step {i:4d}: model_loss {a["model_loss"]:.4f} | {b["model_loss"]:.4f} | {c["model_loss"]:.4f}   actor {a["imag_actor_loss"]:.5f} | '
              f'{b["imag_actor_loss"]:.5f} | {c["imag_actor_loss"]:.5f}   critic {a["imag_critic_loss"]:.5f} | {b["imag_critic_loss"]:.5f} | '
              f'{c["imag_critic_loss"]:.5f}   (default | fp32 MFMAs | bf16 split on every tile)')
    i, a = res['default'][-1]; _, b = res['fp32'][-1]; _, c = res['split3'][-1]
    print('relative difference after', i, 'steps, default vs fp32 MFMAs:', fmt(rel(a, b)))
    print('relative difference after', i, 'steps, bf16 split vs fp32 MFMAs:', fmt(rel(c, b)), flush=True)
    finals.append((a, b, c))
if S > 1:
    print(f'== {S} seeds, after {N} steps: relative difference to the fp32-MFMA run of the SAME seed (median / max over seeds) and the spread of the '
          f'fp32-MFMA runs ACROSS seeds (std / |mean|)')
    print(f'{"metric":24s} {"default: median":>16s} {"max":>9s} {"bf16 split: median":>19s} {"max":>9s} {"across seeds (fp32)":>20s}')
    for k in keys[:9]:
        da = sorted(rel(a, b)[k] for a, b, c in finals); dc = sorted(rel(c, b)[k] for a, b, c in finals)
        vals = torch.tensor([b[k] for a, b, c in finals], dtype=torch.float64)
        print(f'{k:24s} {da[len(da) // 2]:16.1e} {da[-1]:9.1e} {dc[len(dc) // 2]:19.1e} {dc[-1]:9.1e} {(vals.std() / vals.mean().abs().clamp_min(1e-12)).item():20.1e}')
