"""The convolution weight-gradient kernel (gemm_planes_tn_kernel<true>: patches gathered by the DMA) per layer shape, against the same
(NI, NJ, M) product on a MATERIALISED operand (gemm_planes_tn_kernel<false>): microseconds, and microseconds per tile-stage per workgroup
(time x workgroups / (tiles x M / 64)) -- the MFMA stream of a 128 x 128 x 64 stage is ~1.2-1.45 us.  Graph-timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes, ops_conv_planes as ocp
from small_m import graph_time
torch.manual_seed(0)
# (Nimg, H, W, C, k, NI): image of the gathered operand, kernel, columns of the other operand
SH = [(1024, 31, 31, 48, 4, 96), (1024, 14, 14, 96, 4, 192), (1024, 6, 6, 192, 4, 384), (1024, 30, 30, 48, 6, 96), (1024, 13, 13, 96, 5, 192),
      (1024, 63, 63, 48, 4, 96), (1024, 30, 30, 96, 4, 192), (1024, 29, 29, 96, 5, 192), (1024, 63, 63, 48, 6, 96), (1024, 13, 13, 192, 5, 384)]
if len(sys.argv) > 1:
    SH = SH[:int(sys.argv[1])]
for (Nimg, H, W, C, k, NI) in SH:
    Ho, Wo = (H - k) // 2 + 1, (W - k) // 2 + 1
    M = Nimg * Ho * Wo
    if M % 64:
        continue
    NJ = k * k * C
    img = ocp._uniform_split(torch.randn(Nimg * H * W, C, device='cuda'))
    A = planes.split(torch.randn(M, NI, device='cuda'))
    out = torch.empty(NI, NJ, device='cuda')
    tc = min(graph_time(lambda: ocp._gemm_tn_conv(A, img, Nimg, H, W, C, k, out, NJ, NI, M), n=10, reps=5) for _ in range(3))
    Bm = planes.split(torch.randn(M, NJ, device='cuda'))
    tp = min(graph_time(lambda: planes.gemm_tn(A, Bm, out, NJ, NI, NJ, M), n=10, reps=5) for _ in range(3))
    del Bm
    tiles = -(-NI // 128) * -(-NJ // 128)
    ts = tiles * (M // 64)
    from genrl_amd._lib import lib
    print(f'image {Nimg}x{H}x{W}x{C} k={k}: NI={NI:4d} NJ={NJ:5d} M={M:7d} tiles={tiles:3d}: gathered {tc:8.1f} us   materialised {tp:8.1f} us   '
          f'(gathered: {2.0 * NI * NJ * M / tc / 1e6:5.0f} TF/s)', flush=True)
