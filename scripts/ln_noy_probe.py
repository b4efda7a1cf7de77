"""LayerNorm + SiLU forward emitting planes (genrl_ln_act_fwd_h2), 1024 / 16384 rows x 1024: with and without the fp32 copy of the output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes
from genrl_amd._lib import lib, check
from small_m import graph_time
for M in (1024, 16384):
    N = 1024
    nset = 8
    x = [torch.randn(M, N, device='cuda') for _ in range(nset)]
    y = torch.empty(M, N, device='cuda')
    g, b = torch.ones(N, device='cuda'), torch.zeros(N, device='cuda')
    mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    P = planes.Planes(M, N, 'cuda')
    st = lambda: torch.cuda.current_stream().cuda_stream
    for noy in (False, True):
        def run():
            for xi in x:
                check(lib().genrl_ln_act_fwd_h2(xi.data_ptr(), N, g.data_ptr(), b.data_ptr(), None if noy else y.data_ptr(), N, mean.data_ptr(), rstd.data_ptr(),
                                                M, N, 1e-5, 1, P.ptr(0), P.ld, P.plane, P.inv_ptr(0), st()), 'ln')
        t = min(graph_time(run, n=6) / nset for _ in range(3))
        print(f'{M} x {N} LayerNorm + SiLU -> planes, fp32 output {"skipped" if noy else "written"}: {t:6.2f} us', flush=True)
