"""fp32 MFMA vs bf16x3-split (fp32-accurate) vs plain bf16 operands on the step's main shapes (event-timed,
back-to-back), with the maximum error of each mode against a float64 product."""
import sys; sys.path.insert(0, '.')
import torch
from genrl_amd import ops
def t(M, N, K, lay, A, B, C):
    a = (K, 1) if lay[0] == 'k' else (1, M); b = (K, 1) if lay[1] == 'k' else (1, N)
    run = lambda: ops.sgemm(A, a[0], a[1], B, b[0], b[1], C, N, None, M, N, K)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 20
for shp in [(1024, 1024, 1024, 'kk'), (1024, 3072, 1024, 'kk'), (16384, 1024, 1024, 'kk'), (16384, 1024, 1024, 'kr'),
            (1024, 1024, 16384, 'rr'), (1024, 1024, 1024, 'rr'), (1024, 1024, 1024, 'kr')]:
    M, N, K, lay = shp
    Am = torch.randn(M, K, device='cuda'); Bm = torch.randn(N, K, device='cuda'); C = torch.empty(M, N, device='cuda')
    A = Am if lay[0] == 'k' else Am.T.contiguous(); B = Bm if lay[1] == 'k' else Bm.T.contiguous()
    ref = (Am[:256].double() @ Bm.double().T)
    out = []
    for mode in ('f32', 'bf16x3', 'bf16'):
        ops.set_gemm_precision(mode)
        us = t(M, N, K, lay, A, B, C)
        err = ((C[:256].double() - ref).abs().max() / ref.abs().max()).item()
        out.append(f'{mode} {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF/s err {err:.1e}')
    ops.set_gemm_precision('f32')
    print(shp, ' | '.join(out))
