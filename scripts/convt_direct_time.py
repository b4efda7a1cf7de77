"""graph-timed cost of the decoder's last layer forward: genrl_convt_small_co_fwd against GEMM -> col2im (scripts, GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import ops
from small_m import graph_time
N, Hi, Ci, Co, k = 1024, 30, 48, 3, 6
x = torch.randn(N, Hi, Hi, Ci, device='cuda'); W = torch.randn(Ci, Co, k, k, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
for direct in (True, False):
    ops.CONVT_DIRECT = direct
    with torch.no_grad():
        t = graph_time(lambda: ops.convT2d_s2(x, W, b, out_nchw=True), n=10, reps=10)
    print(f'direct={direct}: {t:.1f} us per call (incl. the weight permutation launch)')
