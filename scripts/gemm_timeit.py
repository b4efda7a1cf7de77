"""Event-timed average of one GEMM shape (20 back-to-back launches incl. any split-K reduce)."""
import sys, os
sys.path.insert(0, '.')
import torch
from genrl_amd import ops
M, N, K = [int(x) for x in sys.argv[1:4]]; mode = sys.argv[4] if len(sys.argv) > 4 else 'kk'
A = torch.randn(M * K, device='cuda'); B = torch.randn(N * K, device='cuda'); C = torch.empty(M, N, device='cuda')
a = (K, 1) if mode[0] == 'k' else (1, M); b = (K, 1) if mode[1] == 'k' else (1, N)
run = lambda: ops.sgemm(A, a[0], a[1], B, b[0], b[1], C, N, None, M, N, K)
for _ in range(3): run()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / reps
print(f'{M:7d} {N:6d} {K:7d} {mode} force={os.environ.get("GENRL_GEMM_FORCE", "-"):8s} {us:9.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s', flush=True)
