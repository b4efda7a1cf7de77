"""Crash / finiteness check of one eager training step at the sizes of the other BASELINE configs (c3-sized batch, c4 128x128
5-layer convs): exercises the launch planner and the special-shape GEMM kernels away from the headline shapes."""
import sys, contextlib; sys.path.insert(0, '.')
import torch, numpy as np
import bench
from genrl_amd import config
def run(B, T, A, img, steps=2, **over):
    cfg = config.default_cfg(B, T, device='cuda:0', overlap_detached=True, **over)
    with contextlib.redirect_stdout(sys.stderr):
        ag = config.make_agent(cfg, act_dim=A, img=img)
    ag.wm.viclip_model = bench.TextStub()
    g = np.random.default_rng(0)
    batch = dict(observation=torch.from_numpy(g.integers(0, 256, (B, T, 3, img, img), dtype=np.uint8)),
                 action=torch.from_numpy(g.standard_normal((B, T, A), dtype=np.float32)),
                 reward=torch.zeros(B, T, 1), discount=torch.ones(B, T, 1),
                 is_first=torch.zeros(B, T, dtype=torch.bool), is_last=torch.zeros(B, T, dtype=torch.bool),
                 is_terminal=torch.zeros(B, T, dtype=torch.bool), clip_video=torch.from_numpy(g.standard_normal((B, T, 512), dtype=np.float32)))
    batch['is_first'][:, 0] = True
    batch = {k: v.cuda() for k, v in batch.items()}
    import time
    for i in range(steps + 1):
        if i == 1: torch.cuda.synchronize(); t0 = time.perf_counter()
        m = bench.one_step(ag, batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    vals = {k: float(v) for k, v in m.items()}
    assert all(np.isfinite(v) for v in vals.values()), vals
    print(f'B{B} T{T} A{A} img{img}: {ms:.1f} ms/step (eager), model_loss {vals["model_loss"]:.2f}, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
run(64, 48, 6, 64)       # c3-sized batch on one GPU (GenRL needs T % 8 == 0)
run(32, 32, 9, 128, encoder=dict(cnn_depth=48, cnn_kernels=[4, 4, 4, 4, 4]), decoder=dict(cnn_depth=48, cnn_kernels=[5, 5, 5, 6, 6]))   # c4 kitchen 128x128, 5-layer convs
run(8, 48, 10, 64)
