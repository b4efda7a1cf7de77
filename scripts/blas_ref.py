"""What the vendor fp32 GEMM (torch.matmul -> rocBLAS/hipBLASLt) reaches on this workload's shapes: a yardstick
for genrl_sgemm, not part of the product."""
import torch, sys
sys.path.insert(0, '.')
from genrl_amd import ops
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [(1024, 1024, 1024), (1024, 3072, 1024), (1024, 1024, 3072), (16384, 1024, 1024), (17408, 1536, 1024), (4096, 1024, 1024),
          (128, 1024, 1024), (32, 3072, 1024), (173056, 96, 1728), (200704, 96, 768)]
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps
for M, N, K in shapes:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
    tb = t(lambda: torch.matmul(A, W.t(), out=C))
    tm = t(lambda: ops.sgemm(A, K, 1, W, K, 1, C, N, None, M, N, K))
    fl = 2.0 * M * N * K
    print(f'{M:7d} {N:5d} {K:5d}  vendor {tb:8.1f} us {fl/tb/1e6:6.1f} TF/s   genrl {tm:8.1f} us {fl/tm/1e6:6.1f} TF/s', flush=True)
