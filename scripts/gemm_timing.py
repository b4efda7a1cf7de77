"""Per-phase cycle breakdown of sgemm_kernel (needs a -DGENRL_DBG_TIMING build: GENRL_HIP_SO=/tmp/lib_t.so)."""
import sys, ctypes
sys.path.insert(0, '.')
import numpy as np
import torch
from genrl_amd import ops, _lib
M, N, K = [int(x) for x in sys.argv[1:4]]; mode = sys.argv[4] if len(sys.argv) > 4 else 'kk'
A = torch.randn(M * K, device='cuda'); B = torch.randn(N * K, device='cuda'); C = torch.empty(M, N, device='cuda')
a = (K, 1) if mode[0] == 'k' else (1, M); b = (K, 1) if mode[1] == 'k' else (1, N)
L = _lib.lib()
for _ in range(4): ops.sgemm(A, a[0], a[1], B, b[0], b[1], C, N, None, M, N, K)
torch.cuda.synchronize()
tiles = ((M + 63) // 64) * ((N + 63) // 64)
waves = min(tiles * 16, 65536)
out = np.zeros((waves, 6), np.uint64)
L.genrl_dbg_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.genrl_dbg_read(out.ctypes.data, waves)
names = ['fetch-issue', 'compute', 'stage(+vmcnt)', 'barrier', 'prologue', 'epilogue']
o = out.astype(np.float64)
tot = o.sum(1)
print(f'{M}x{N}x{K} {mode}: per-wave cycles over {waves} waves: total mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}')
for i, n_ in enumerate(names):
    print(f'  {n_:14s} mean {o[:, i].mean():9.0f}  {100.0 * o[:, i].sum() / tot.sum():5.1f}%   min {o[:, i].min():8.0f} max {o[:, i].max():8.0f}')
