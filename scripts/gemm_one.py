import sys
sys.path.insert(0, '.')
import torch
from genrl_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]; mode = sys.argv[4] if len(sys.argv) > 4 else 'kk'
A = torch.randn(M * K, device='cuda'); B = torch.randn(N * K, device='cuda'); C = torch.empty(M, N, device='cuda')
a = (K, 1) if mode[0] == 'k' else (1, M); b = (K, 1) if mode[1] == 'k' else (1, N)
for _ in range(5): ops.sgemm(A, a[0], a[1], B, b[0], b[1], C, N, None, M, N, K)
torch.cuda.synchronize()
