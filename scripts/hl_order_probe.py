"""gemm_planes_hl_kernel<false> on 16384 x 1024 x 1024 (and 16384 x 1536 x 1024) under the tile-order experiments (GENRL_XCD_M, GENRL_HL_ORDER):
graph-timed, rotating over 4 operand sets.  With --once: 8 plain launches (for a rocprofv3 --pmc pass)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes
from small_m import graph_time
torch.manual_seed(0)
once = '--once' in sys.argv
for (M, N, K) in [(16384, 1024, 1024)] + ([] if once else [(16384, 1536, 1024), (16384, 1024, 2048)]):
    nset = 4
    A = [planes.split(torch.randn(M, K, device='cuda')) for _ in range(nset)]
    W = [planes.split(torch.randn(N, K, device='cuda') * 0.05) for _ in range(nset)]
    C = [torch.empty(M, N, device='cuda') for _ in range(nset)]
    if once:
        for r in range(2):
            for i in range(nset):
                planes.gemm(A[i], W[i], C[i], N, None, M, N)
        torch.cuda.synchronize()
        continue

    def rot():
        for i in range(nset):
            planes.gemm(A[i], W[i], C[i], N, None, M, N)
    t = min(graph_time(rot, n=4) / nset for _ in range(3))
    print(f'XCD_M={os.environ.get("GENRL_XCD_M", "auto"):4s} ORDER={os.environ.get("GENRL_HL_ORDER", "0")}  {M}x{N}x{K}: {t:7.2f} us', flush=True)
