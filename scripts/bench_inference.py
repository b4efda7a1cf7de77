"""Latency of the inference rows at full dims on one MI355X:
act() per environment step (batch 1: encoder + obs_step + actor) and report() (video_pred,
text-to-video, video_clip_pred decode) per call.  python scripts/bench_inference.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from genrl_amd import config
from genrl_amd.tools import genrl_utils as GU
from bench import synth_batch, TextStub

torch.manual_seed(0)
B, T = 8, 32
cfg = config.default_cfg(B, T, device='cuda')
cfg['additional_report_fns'] = ['report_text2video']
ag = config.make_agent(cfg)
ag.wm.viclip_model = TextStub()
GU.DOMAIN2PREDICATES['stickman'] = [f'behaviour {i}' for i in range(12)]
batch_np = synth_batch(B, T)
batch = {k: torch.from_numpy(v).cuda() for k, v in batch_np.items()}


def timed(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

state = [None]
def act_step(eval_mode=True):
    t = np.random.randint(T)
    obs = {k: v[0, t] for k, v in batch_np.items() if k != 'action'}
    _, state[0] = ag.act(obs, None, 0, eval_mode, state[0])

out = {'act_eval_ms_per_env_step': timed(lambda: act_step(True), 200),
       'act_explore_ms_per_env_step': timed(lambda: act_step(False), 200),
       'report_ms_per_call_B8_T32': timed(lambda: ag.report(batch), 10)}
print(json.dumps(out))
