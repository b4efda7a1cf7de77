"""The plane GEMM with operands COLD in the per-XCD L2s, as it meets them inside the step: a rotation over `nset` distinct
(A, W, C) sets (activations written by a producer launch right before the product, weights last read nset launches ago) against
the same launch repeated back to back (operands L2-resident).  Variants of genrl_planes_variant: ring depth (2, 4; five stages no longer fit beside the epilogue factors), L2 prefetch
distance -- compiled in by scripts/build_exp.sh (the shipped library holds 'ring 3' only and ignores the switch).
GPU box only: scripts/build_exp.sh && GENRL_HIP_SO=gpurun_exp.so python scripts/cold_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd._lib import lib
from planes_bench import split2, gemm2
from x3_bench import dev
from small_m import graph_time

VARIANTS = {0: 'ring 3', 1: 'ring 3 + prefetch 3 (128-tile: ring 2 + prefetch 2)', 2: 'ring 3 + prefetch 6 (128: +4)', 3: 'ring 3 + prefetch 10 (128: +4)',
            4: 'ring 4', 6: 'ring 2 + prefetch 6 (128: +2)'}


def run(M, N, K, nset, with_producer):
    torch.manual_seed(0)
    As = [torch.randn(M, K, device=dev) for _ in range(nset)]
    Ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(nset)]
    Cs = [torch.empty(M, N, device=dev) for _ in range(nset)]
    ap = [split2(a) for a in As]
    wp = [split2(w) for w in Ws]
    src = [a[0].clone() for a in ap]
    ref = (As[1].double() @ Ws[1].double().t())
    out = {}
    for v, name in VARIANTS.items():
        lib().genrl_planes_variant(v)
        gemm2(ap[1], wp[1], Cs[1])
        err = (Cs[1].double() - ref).abs().max().item() / ref.abs().mean().item()
        hot = graph_time(lambda: gemm2(ap[0], wp[0], Cs[0]))

        def cold():
            for i in range(nset):
                if with_producer:        # the activation planes are re-written by another launch right before (a copy: same bytes)
                    ap[i][0].copy_(src[i])
                gemm2(ap[i], wp[i], Cs[i])
        def prod_only():
            for i in range(nset):
                ap[i][0].copy_(src[i])
        t = graph_time(cold)
        if with_producer:
            t -= graph_time(prod_only)
        out[v] = (hot, t / nset, err)
        print(f'  {name:58s} hot {hot:7.1f} us   cold {t / nset:7.1f} us   err {err:.1e}', flush=True)
    lib().genrl_planes_variant(0)
    return out


if __name__ == '__main__':
    for (M, N, K, nset) in [(1024, 1024, 1024, 16), (1024, 3072, 1024, 8), (1024, 1024, 2048, 8), (16384, 1024, 1024, 3)]:
        for prod in (False, True):
            print(f'{M}x{N}x{K}, {nset} operand sets, producer launch before each product: {prod}', flush=True)
            run(M, N, K, nset, prod)
