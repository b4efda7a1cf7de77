"""graph-timed cost of the decoder's last layer, forward and backward: the direct fp32-MFMA gather kernels (genrl_convt_small_co_fwd / _bwd)
against GEMM -> col2im / im2col + two GEMMs (GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import ops
from small_m import graph_time
N, Hi, Ci, Co, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 30, 48, 3, 6
x = torch.randn(N, Hi, Hi, Ci, device='cuda', requires_grad=True); W = (torch.randn(Ci, Co, k, k, device='cuda') * 0.05).requires_grad_(True)
b = torch.randn(Co, device='cuda', requires_grad=True)
gy = torch.randn(N, Co, 64, 64, device='cuda')
for direct in ((True,) if len(sys.argv) > 2 else (True, False)):
    ops.CONVT_DIRECT = direct
    with torch.no_grad():
        tf = graph_time(lambda: ops.convT2d_s2(x, W, b, out_nchw=True), n=10, reps=10)

    def fb():
        y = ops.convT2d_s2(x, W, b, out_nchw=True)
        torch.autograd.grad(y, (x, W, b), gy)
    tfb = graph_time(fb, n=5, reps=10)
    print(f'direct={direct}: forward {tf:.1f} us, forward + backward {tfb:.1f} us (incl. weight permutations and the bias-gradient reduction)')
