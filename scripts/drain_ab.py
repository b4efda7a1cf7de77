"""A/B of one library build: graph-timed plane products on the step's shapes (64x64 and 128x128 tiles), back to back and rotating over
16 operand sets.  GENRL_HIP_SO=<build> python scripts/drain_ab.py <label>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes
from small_m import graph_time

label = sys.argv[1] if len(sys.argv) > 1 else 'build'
torch.manual_seed(0)
for (M, N, K) in [(1024, 1024, 1024), (1024, 1024, 3072), (1024, 3072, 1024), (16384, 1024, 1024), (16384, 1536, 1024)]:
    nset = 16 if M <= 1024 else 4
    A = [planes.split(torch.randn(M, K, device='cuda')) for _ in range(nset)]
    W = [planes.split(torch.randn(N, K, device='cuda') * 0.05) for _ in range(nset)]
    C = [torch.empty(M, N, device='cuda') for _ in range(nset)]
    bias = torch.randn(N, device='cuda')
    ts = []
    for _ in range(3):
        hot = graph_time(lambda: planes.gemm(A[0], W[0], C[0], N, bias, M, N))

        def rot():
            for i in range(nset):
                planes.gemm(A[i], W[i], C[i], N, bias, M, N)
        cold = graph_time(rot, n=4) / nset
        ts.append((hot, cold))
    print(f'{label:14s} {M}x{N}x{K}: back to back ' + ' '.join(f'{h:7.2f}' for h, _ in ts) + '   rotating ' + ' '.join(f'{c:7.2f}' for _, c in ts), flush=True)
