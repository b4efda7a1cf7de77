"""64x64 plane tile: one workgroup per CU on a three-stage ring (shipped) against two co-resident workgroups on a two-stage ring
(GENRL_PLANES_2PER=1), graph-timed, back to back and rotating over operand sets.  python scripts/two_per_cu.py"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for v in ('0', '1'):
        subprocess.run([sys.executable, __file__, v], env=dict(os.environ, GENRL_PLANES_2PER=v))
    sys.exit(0)
import torch
from genrl_amd import planes
from small_m import graph_time
label = 'two per CU (NS=2)' if sys.argv[1] == '1' else 'one per CU (NS=3)'
torch.manual_seed(0)
for (M, N, K) in [(3200, 512, 512), (3200, 512, 1024), (3200, 512, 1536), (3200, 1024, 512), (3200, 1536, 1024), (1024, 3072, 2048),
                  (1024, 1536, 1024), (2048, 1024, 1024), (1024, 1024, 1024), (512, 1024, 1024)]:
    nset = 8
    A = [planes.split(torch.randn(M, K, device='cuda')) for _ in range(nset)]
    W = [planes.split(torch.randn(N, K, device='cuda') * 0.05) for _ in range(nset)]
    C = [torch.empty(M, N, device='cuda') for _ in range(nset)]
    bias = torch.randn(N, device='cuda')
    planes_force = planes.lib().genrl_planes_force_tile(1)
    hot = min(graph_time(lambda: planes.gemm(A[0], W[0], C[0], N, bias, M, N)) for _ in range(3))

    def rot():
        for i in range(nset):
            planes.gemm(A[i], W[i], C[i], N, bias, M, N)
    cold = min(graph_time(rot, n=6) / nset for _ in range(3))
    planes.lib().genrl_planes_force_tile(planes_force)
    tiles = -(-M // 64) * -(-N // 64)
    print(f'{label:20s} {M:5d}x{N:4d}x{K:4d} ({tiles:4d} tiles): back to back {hot:7.2f} us   rotating {cold:7.2f} us', flush=True)
