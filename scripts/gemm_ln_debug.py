"""one fused Dense -> LayerNorm launch, then a dump of its exchange records (debugging aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes
from genrl_amd._lib import lib
for (M, N, K) in [(64, 64, 64), (64, 1024, 256), (128, 1024, 256), (1024, 1024, 1024)]:
    A = planes.split(torch.randn(M, K, device='cuda')); B = planes.split(torch.randn(N, K, device='cuda') / K ** .5)
    C = torch.empty(M, N, device='cuda'); y = torch.empty(M, N, device='cuda')
    g = torch.ones(N, device='cuda'); b = torch.zeros(N, device='cuda'); out = planes.Planes(M, N, 'cuda')
    print('ok?', planes.gemm_ln_ok(M, N))
    for rep in range(2):
        planes.gemm_ln(A, B, C, None, M, N, g, b, 1e-3, out, 0, y=y)
        torch.cuda.synchronize()
        sync, part = planes._ln_ws[0]
        off = ((part.data_ptr() + 15) // 16 * 16 - part.data_ptr()) // 4
        tn = N // 64
        slab = part[off + (tn - 1) * 65536:][: (M // 64) * tn * 256].view(M // 64, tn, 64, 4)
        tags = slab[..., 1].view(torch.int32)
        print(f'{M}x{N}x{K} rep {rep}: fail word {int(sync[0])}; tags (row block 0, every tile, rows 0 / 63): {tags[0, :, 0].tolist()} {tags[0, :, 63].tolist()}; tag2 {slab[0, :, 0, 3].view(torch.int32).tolist()}')
        print('   mean of tile 0 row 0:', float(slab[0, 0, 0, 0]), 'expected', float(C[0, :64].mean()))
        sync.zero_()
