"""What the K loop of the 64x64 plane tile is made of (round 4, verdict item 1): the shipped kernel against ablation builds of the SAME
kernel (scripts/build_abl.sh: PLANES_ABL 2 = no DMA in the loop, 3 = no fragment reads, 4 = neither, 1 = no MFMAs; results are wrong by
construction), graph-timed on the rollout's shapes, back to back and over 16 rotating operand sets.  'no fragment reads' is an UPPER
bound on what taking the weight operand's fragment reads off the LDS port could buy (it removes BOTH operands' reads).
GPU box only:  for a in 0 1 2 3 4; do GENRL_HIP_SO=$PWD/gpurun_abl$a.so python scripts/abl64.py $a; done   (0 = libgenrl_hip.so)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes
from small_m import graph_time

NAMES = {'0': 'shipped', '1': 'no MFMAs', '2': 'no DMA in the loop', '3': 'no fragment reads', '4': 'neither DMA nor reads', '5': 'no barrier either'}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else '0'
    torch.manual_seed(0)
    for (M, N, K) in [(1024, 1024, 1024), (1024, 1024, 2048), (1024, 1024, 3072)]:
        nset = 16
        A = [planes.split(torch.randn(M, K, device='cuda')) for _ in range(nset)]
        W = [planes.split(torch.randn(N, K, device='cuda') * 0.05) for _ in range(nset)]
        C = [torch.empty(M, N, device='cuda') for _ in range(nset)]
        hot = graph_time(lambda: planes.gemm(A[0], W[0], C[0], N, None, M, N))

        def rot():
            for i in range(nset):
                planes.gemm(A[i], W[i], C[i], N, None, M, N)
        cold = graph_time(rot, n=4) / nset
        print(f'{NAMES.get(tag, tag):24s} {M}x{N}x{K}: back to back {hot:6.2f} us   rotating over {nset} operand sets {cold:6.2f} us', flush=True)


if __name__ == '__main__':
    main()
