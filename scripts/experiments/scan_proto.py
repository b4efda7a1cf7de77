"""N1 prototype measurement: the forward GRU recurrence of EnsembleRSSM.observe (T = 32 steps, D = 1024, I = 1024) as
 (a) the product's per-step launches: skinny GEMV (h W_h^T, W_h streamed from the fabric every step) + LayerNorm/gate block,
 (b) the state-resident persistent kernel with two grid barriers per step (csrc/scan_coop.hip, variant 2),
 (c) the same with ONE barrier per step and redundant gate evaluation (variant 1, B <= 8),
graph-timed (the x-projection for all T is inside every variant, as in ops._GRUSeq.forward).  python scripts/scan_proto.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import ops
from small_m import graph_time

from genrl_amd._lib import lib, check
T, I, D = 32, 1024, 1024
ws = torch.empty(4096, device='cuda'); wsp = (ws.data_ptr() + 255) // 256 * 256
for G in (256, 64):
    for n in (64, 256):
        tb = graph_time(lambda: check(lib().genrl_grid_barrier_bench(wsp, n, G, torch.cuda.current_stream().cuda_stream), 'bar'), n=3, reps=10)
        t0 = graph_time(lambda: check(lib().genrl_grid_barrier_bench(wsp, 0, G, torch.cuda.current_stream().cuda_stream), 'bar'), n=3, reps=10)
        print(f'grid barrier (XCD-hierarchical, {G} workgroups of 256 threads): {(tb - t0) / n:5.2f} us per barrier ({n} barriers, launch + setup {t0:.1f} us)', flush=True)
print(f'GRU scan forward, T={T}, D={D}, I={I}: microseconds per SEQUENCE (per step in brackets), graph-timed, incl. the batched x-projection')
for B in (32, 16, 8, 4):
    g = torch.Generator(device='cuda').manual_seed(B)
    x = torch.randn(T, B, I, device='cuda', generator=g)
    h0 = torch.randn(B, D, device='cuda', generator=g) * 0.5
    W = torch.randn(3 * D, I + D, device='cuda', generator=g) / (I + D) ** 0.5
    gamma = 1.0 + 0.1 * torch.randn(3 * D, device='cuda', generator=g); beta = 0.1 * torch.randn(3 * D, device='cuda', generator=g)
    mask = torch.ones(T, B, device='cuda'); mask[0] = 0
    res = {}
    outs = {}
    for variant in (0, 2, 1):
        if variant == 1 and B > 8:
            continue
        os.environ['GENRL_SCAN_COOP'] = str(variant)
        with torch.no_grad():
            outs[variant] = ops.gru_seq(x, mask, h0, W, gamma, beta).clone()
            res[variant] = graph_time(lambda: ops.gru_seq(x, mask, h0, W, gamma, beta), n=5, reps=10)
            xproj = graph_time(lambda: ops.sgemm(x.reshape(T * B, I), I, 1, W, I + D, 1, torch.empty(T * B, 3 * D, device='cuda'), 3 * D, None, T * B, 3 * D, I), n=5, reps=10)
    os.environ.pop('GENRL_SCAN_COOP', None)
    line = f'B={B:2d}: x-projection alone {xproj:7.1f} | per-step launches {res[0]:7.1f} ({(res[0] - xproj) / T:5.1f}/step)'
    line += f' | persistent, 2 barriers {res[2]:7.1f} ({(res[2] - xproj) / T:5.1f}/step) maxdiff {(outs[2] - outs[0]).abs().max().item():.1e}'
    if 1 in res:
        line += f' | persistent, 1 barrier {res[1]:7.1f} ({(res[1] - xproj) / T:5.1f}/step) maxdiff {(outs[1] - outs[0]).abs().max().item():.1e}'
    print(line, flush=True)
