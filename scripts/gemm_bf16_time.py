"""fp32 vs bf16-operand MFMA mode of the GEMM engine on the step's main shapes (event-timed, back-to-back)."""
import sys; sys.path.insert(0, '.')
import torch
from genrl_amd import ops
def t(M, N, K, lay):
    A = torch.randn(M * K, device='cuda'); B = torch.randn(N * K, device='cuda'); C = torch.empty(M, N, device='cuda')
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
            (1024, 1024, 16384, 'rr'), (1024, 1024, 1024, 'rr')]:
    ops.set_gemm_precision('f32'); f = t(*shp)
    ops.set_gemm_precision('bf16'); h = t(*shp)
    ops.set_gemm_precision('f32')
    fl = 2.0 * shp[0] * shp[1] * shp[2] / 1e6
    print(f'{shp}: f32 {f:7.1f} us ({fl / f:6.1f} TF/s)   bf16 {h:7.1f} us ({fl / h:6.1f} TF/s)')
