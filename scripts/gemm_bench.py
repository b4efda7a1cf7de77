"""Micro-benchmark of genrl_sgemm on the shapes of the GenRL step (HIP events, random data)."""
import sys, json
sys.path.insert(0, '.')
import torch
from genrl_amd import ops
shapes = [  # (M, N, K, mode)
    (1024, 1024, 1024, 'kk'), (1024, 1024, 1024, 'kr'), (1024, 1024, 1024, 'rr'), (1024, 3072, 1024, 'kk'),
    (1024, 1024, 3072, 'kr'), (1024, 1024, 16384, 'rr'), (17408, 1024, 1024, 'kk'), (17408, 1024, 1024, 'kr'),
    (32, 3072, 1024, 'kk'), (32, 1024, 3072, 'kr'), (1024, 1024, 2048, 'kk'), (1024, 1536, 1024, 'kk'),
    (96, 1728, 173056, 'rr'), (173056, 96, 1728, 'kk'), (173056, 1728, 96, 'kr'), (48, 48, 984064, 'rr'),
    (921600, 48, 108, 'kk'), (10, 1024, 1024, 'rr'), (1024, 10, 1024, 'kk'), (255, 1024, 16384, 'rr'),
    (4096, 4096, 4096, 'kk'),
]
def run(M, N, K, mode, iters=10):
    A = torch.randn(M * K, device='cuda'); B = torch.randn(N * K, device='cuda'); C = torch.empty(M, N, device='cuda')
    a = (K, 1) if mode[0] == 'k' else (1, M)
    b = (K, 1) if mode[1] == 'k' else (1, N)
    for _ in range(2): ops.sgemm(A, a[0], a[1], B, b[0], b[1], C, N, None, M, N, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.sgemm(A, a[0], a[1], B, b[0], b[1], C, N, None, M, N, K)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    return us, 2.0 * M * N * K / us / 1e6
if len(sys.argv) > 1:   # correctness spot check
    M, N, K = 200, 136, 1111
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
    ops.sgemm(A, K, 1, B, K, 1, C, N, None, M, N, K); print('maxerr', (C - A @ B.T).abs().max().item())
for s in (shapes if __name__ == "__main__" else []):
    us, tf = run(*s)
    print(f'{s[0]:7d} {s[1]:5d} {s[2]:7d} {s[3]}  {us:9.1f} us  {tf:7.2f} TF/s')
