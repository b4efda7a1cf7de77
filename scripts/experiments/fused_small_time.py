"""graph-timed cost of one few-row layer: genrl_small_fused (LayerNorm in the loader) against the weight-streaming / tile product + the
LayerNorm row kernel it replaces.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import ops
from genrl_amd._lib import lib, check
from small_m import graph_time
st = lambda: torch.cuda.current_stream().cuda_stream
for M in (128, 256):
    for (N, K0, K1) in [(1024, 1024, 0), (3072, 1024, 1024), (1024, 1024, 1024)]:
        a0 = torch.randn(M, K0, device='cuda'); w0 = torch.randn(N, K0 + K1, device='cuda') * 0.03
        a1 = torch.randn(M, K1, device='cuda') if K1 else None
        g = torch.ones(K0, device='cuda'); be = torch.zeros(K0, device='cuda'); b = torch.zeros(N, device='cuda')
        stats = torch.zeros(K0 // 16, M, 2, device='cuda'); stats[..., 1] = 16.0
        C = torch.empty(M, N, device='cuda'); so = torch.empty(N // 16, M, 2, device='cuda')
        y = torch.empty(M, K0, device='cuda'); mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
        seg1 = (a1.data_ptr(), K1, w0.data_ptr() + 4 * K0, K0 + K1, K1) if K1 else None

        def fused_ln():
            ops.small_fused(a0.data_ptr(), K0, w0.data_ptr(), K0 + K1, K0, C.data_ptr(), N, M, N, ln=(stats.data_ptr(), K0 // 16, g, be, 1e-3), seg1=seg1, bias=b,
                            stats_out=so.data_ptr())

        def fused_plain():
            ops.small_fused(a0.data_ptr(), K0, w0.data_ptr(), K0 + K1, K0, C.data_ptr(), N, M, N, seg1=seg1, bias=b, stats_out=so.data_ptr())

        def unfused():
            check(lib().genrl_ln_act_fwd(a0.data_ptr(), K0, g.data_ptr(), be.data_ptr(), y.data_ptr(), K0, mean.data_ptr(), rstd.data_ptr(), M, K0, 1e-3, 1, st()), 'ln')
            ops.sgemm(y, K0, 1, w0, K0 + K1, 1, C, N, b, M, N, K0)
            if K1:
                ops.sgemm(a1, K1, 1, w0, K0 + K1, 1, C, N, None, M, N, K1, accumulate=True, b_off=K0)
        print(f'M={M} N={N} K={K0}+{K1}: fused with LN {graph_time(fused_ln):6.1f} us | fused, plain rows {graph_time(fused_plain):6.1f} us | '
              f'LayerNorm kernel + product(s) {graph_time(unfused):6.1f} us', flush=True)
