import sys
sys.path.insert(0, '.')
import torch
from scripts.gemm_bench import run
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    us, tf = run(1024, 1024, K, 'kk', iters=20)
    print(f'K={K:5d} {us:8.1f} us {tf:7.2f} TF/s')
