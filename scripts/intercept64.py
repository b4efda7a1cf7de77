"""The fixed cost per launch of the 64x64 plane tile (round 5, verdict item 3): 1024 x 1024 x K for K = 64 .. 3072 (a direct fit of slope
and intercept) on the shipped kernel, and the shipped kernel against ablation builds of itself (scripts/build_abl.sh: PLANES_ABL 6 =
the launch alone, 7 = launch + prologue (epilogue-factor DMAs, first three stages, first fragment reads), 8 = launch + prologue +
epilogue (C tile), no K loop).  Graph-timed, back to back and over 16 rotating operand sets.
GPU box only:  for a in 0 6 7 8; do GENRL_HIP_SO=$PWD/gpurun_abl$a.so python scripts/intercept64.py $a; done   (0: libgenrl_hip.so)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes
from small_m import graph_time

NAMES = {'0': 'shipped', '6': 'launch alone', '7': 'launch + prologue', '8': 'launch + prologue + epilogue', '9': 'launch + prologue + epilogue w/o stores',
         '10': 'launch + prologue + epilogue, nt stores', 'nt': 'shipped kernel, nt stores'}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else '0'
    torch.manual_seed(0)
    M = N = 1024
    Ks = [64, 128, 256, 512, 1024, 2048, 3072] if tag == '0' else [1024]
    if tag == 'nt':
        Ks = [64, 1024, 3072]
    rows = []
    for K in Ks:
        nset = 16
        A = [planes.split(torch.randn(M, K, device='cuda')) for _ in range(nset)]
        W = [planes.split(torch.randn(N, K, device='cuda') * 0.05) for _ in range(nset)]
        C = [torch.empty(M, N, device='cuda') for _ in range(nset)]
        bias = torch.randn(N, device='cuda')
        hot = graph_time(lambda: planes.gemm(A[0], W[0], C[0], N, bias, M, N))

        def rot():
            for i in range(nset):
                planes.gemm(A[i], W[i], C[i], N, bias, M, N)
        cold = graph_time(rot, n=4) / nset
        rows.append((K, hot, cold))
        print(f'{NAMES.get(tag, tag):30s} {M}x{N}x{K:5d}: back to back {hot:6.2f} us   rotating over {nset} operand sets {cold:6.2f} us', flush=True)
    if len(rows) > 2:
        import numpy as np
        k = np.array([r[0] / 64.0 for r in rows if r[0] >= 512])
        for name, col in (('back to back', 1), ('rotating', 2)):
            y = np.array([r[col] for r in rows if r[0] >= 512])
            a, b = np.polyfit(k, y, 1)
            print(f'# fit over K >= 512 ({name}): {a:.3f} us per 64-k stage + {b:.2f} us intercept')


if __name__ == '__main__':
    main()
