"""Dense -> LayerNorm -> SiLU: ONE launch (genrl_gemm_h2_ln: XCD-local exchange of the row statistics) against the two it replaces
(genrl_gemm_h2 + genrl_ln_act_fwd_h2), graph-timed, as a chain of `depth` layers (each layer reads the planes the previous one wrote:
the rollout's policy trunk), rotating over distinct weight sets so that no operand is L2-resident from the launch before.
GPU box only:  python scripts/gemm_ln_time.py > profiles/r06_gemm_ln_time.txt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes, ops_planes
from small_m import graph_time

dev = 'cuda'
print('# us per layer (Dense + LayerNorm + SiLU), chain of 4 layers x 4 weight sets, graph-timed; "gemm" = the product alone (same chain, no LayerNorm: its input planes are a fixed tensor)')
for (M, N) in [(1024, 1024), (512, 1024), (256, 1024), (128, 1024), (1024, 512), (2048, 512)]:
    K = N
    torch.manual_seed(0)
    depth, nset = 4, 4
    Ws = [[planes.split(torch.randn(N, K, device=dev) / K ** .5) for _ in range(depth)] for _ in range(nset)]
    bias = torch.zeros(N, device=dev); gamma = torch.ones(N, device=dev); beta = torch.zeros(N, device=dev)
    x0 = planes.split(torch.randn(M, K, device=dev))
    P = [planes.Planes(M, N, dev) for _ in range(depth)]
    C = [torch.empty(M, N, device=dev) for _ in range(depth)]
    Y = [torch.empty(M, N, device=dev) for _ in range(depth)]
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)

    def fused(y=True):
        for s in range(nset):
            a = x0
            for l in range(depth):
                planes.gemm_ln(a, Ws[s][l], C[l], bias, M, N, gamma, beta, 1e-3, P[l], 0, y=Y[l] if y else None, mean=mean, rstd=rstd)
                a = P[l]

    def pair(y=True):
        for s in range(nset):
            a = x0
            for l in range(depth):
                planes.gemm(a, Ws[s][l], C[l], N, bias, M, N)
                ops_planes._ln_fwd(C[l].data_ptr(), gamma, beta, Y[l].data_ptr() if y else None, mean.data_ptr(), rstd.data_ptr(), M, N, 1e-3, P[l], 0)
                a = P[l]

    def gemm_only():
        for s in range(nset):
            for l in range(depth):
                planes.gemm(x0, Ws[s][l], C[l], N, bias, M, N)
    assert planes.gemm_ln_ok(M, N)
    n = depth * nset
    tf, tp, tg = graph_time(fused, n=8) / n, graph_time(pair, n=8) / n, graph_time(gemm_only, n=8) / n
    tf0, tp0 = graph_time(lambda: fused(False), n=8) / n, graph_time(lambda: pair(False), n=8) / n
    planes.check_ln_failure()
    print(f'{M:5d} x {N:4d} x {K:4d}: one launch {tf:6.2f} us   two launches {tp:6.2f} us   gemm {tg:6.2f} us   | planes only (no fp32 y): one {tf0:6.2f}  two {tp0:6.2f}', flush=True)
