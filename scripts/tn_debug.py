import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes, ops
for (M, NI, NJ) in [(512, 128, 128), (1024, 1024, 1024), (16384, 1024, 1024), (2048, 1024, 2048), (1024, 3072, 1024), (4096, 200, 520), (640, 1024, 1536), (17408, 256, 1024), (64, 32, 16), (512, 32, 48)]:
    for rng in (0, 6, 20):
        g = torch.Generator(device='cuda').manual_seed(M + NI + NJ)
        rs = torch.exp2(torch.randint(-rng, rng + 1, (M, 1), device='cuda', generator=g).float())
        dY = torch.randn(M, NI, device='cuda', generator=g) * rs * 1e-3
        X = torch.randn(M, NJ, device='cuda', generator=g) / rs.flip(0) * 3.0
        pa, pb = planes.split(dY), planes.split(X)
        C = torch.zeros(NI, NJ, device='cuda')
        try:
            planes.gemm_tn(pa, pb, C, NJ, NI, NJ, M)
            torch.cuda.synchronize()
        except Exception as e:
            print(M, NI, NJ, rng, 'EXC', type(e).__name__, str(e)[:300], flush=True)
            raise
        ref = dY.double().t() @ X.double()
        asum = dY.double().abs().t() @ X.double().abs()
        err = (C.double() - ref).abs()
        print(f'{M}x{NI}x{NJ} range 2^+-{rng}: max err/mean asum {(err.max() / asum.mean()).item():.2e}  max elementwise err/asum {(err / asum).max().item():.2e}  '
              f'frac > 1e-6 asum: {((err > 1e-6 * asum).double().mean()).item():.2e}', flush=True)
        C2 = torch.zeros(NI, NJ, device='cuda')
        ops.sgemm(dY, 1, NI, X, 1, NJ, C2, NJ, None, NI, NJ, M); torch.cuda.synchronize()
        e2 = (C2.double() - ref).abs()
        print(f'      fp32-operand kernel: max err/mean asum {(e2.max() / asum.mean()).item():.2e}  max elementwise {(e2 / asum).max().item():.2e}', flush=True)
print('tn_debug done')
