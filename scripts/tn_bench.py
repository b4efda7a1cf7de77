"""Weight-gradient product on h2 planes (genrl_gemm_h2_tn, csrc/gemm_planes_tn.hip) against the fp32-operand kernel of the same
product (genrl_sgemm, 'rr' operands: in-register bf16 split on the 128x128 tile / fp32 MFMAs): graph-timed cost and error vs
float64.  GPU box only: python scripts/tn_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes, ops
from genrl_amd._lib import lib
from small_m import graph_time

dev = 'cuda'


def main():
    torch.manual_seed(0)
    shapes = [(16384, 1024, 1024), (17408, 1024, 1024), (16384, 1024, 2048), (1024, 1024, 1024), (2048, 1024, 1024), (4096, 1024, 1024),
              (1024, 3072, 1024), (16384, 256, 1024), (16384, 20, 1024)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
    for (M, NI, NJ) in shapes:
        dY = torch.randn(M, NI, device=dev) * torch.exp2(torch.randint(-6, 7, (M, 1), device=dev).float()) * 1e-2
        X = torch.randn(M, NJ, device=dev)
        pa, pb = planes.split(dY), planes.split(X)
        ref = dY.double().t() @ X.double()
        scale = (dY.double().abs().t() @ X.double().abs()).mean().item()
        C = torch.zeros(NI, NJ, device=dev)
        planes.gemm_tn(pa, pb, C, NJ, NI, NJ, M)
        e1 = (C.double() - ref).abs().max().item() / scale
        t1 = graph_time(lambda: planes.gemm_tn(pa, pb, C, NJ, NI, NJ, M))
        ops.sgemm(dY, 1, NI, X, 1, NJ, C, NJ, None, NI, NJ, M)
        e0 = (C.double() - ref).abs().max().item() / scale
        t0 = graph_time(lambda: ops.sgemm(dY, 1, NI, X, 1, NJ, C, NJ, None, NI, NJ, M))
        fl = 2.0 * M * NI * NJ
        print(f'dW[{NI}x{NJ}] over {M} rows: planes-tn {t1:7.1f} us ({fl / t1 / 1e6:5.0f} TF/s) err {e1:.2e} | fp32-operand kernel {t0:7.1f} us '
              f'({fl / t0 / 1e6:5.0f} TF/s) err {e0:.2e}', flush=True)


if __name__ == '__main__':
    main()
