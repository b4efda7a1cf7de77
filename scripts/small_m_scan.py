"""The observe scan's per-step products at B = 64 / 32 / 8 sequences (Dreamer-v3 widths: 512 / 1024 latent / 1536 gates): the planner's
choice against the weight-streaming kernel forced for every M <= 128 (GENRL_SKINNY_MAX_M=128).  Graph-timed, rotating over 8 weight sets."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    for v in ('planner', 'skinny'):
        env = dict(os.environ)
        if v == 'skinny':
            env['GENRL_SKINNY_MAX_M'] = '128'
        subprocess.run([sys.executable, __file__, v], env=env)
    sys.exit(0)
import torch
from genrl_amd._lib import lib
from small_m import graph_time
dev = 'cuda'
for M in (64, 32, 8):
    for (N, K, kc) in [(512, 1024, True), (1536, 1024, True), (512, 512, True), (1024, 512, True), (512, 1024, False), (512, 512, False),
                       (1024, 1536, False), (1024, 512, False)]:
        nset = 8
        A = torch.randn(M, K, device=dev)
        W = [torch.randn(N, K, device=dev) * 0.05 for _ in range(nset)] if kc else [torch.randn(K, N, device=dev) * 0.05 for _ in range(nset)]
        C = torch.empty(M, N, device=dev)
        ws = torch.empty(max(lib().genrl_sgemm_ws_floats(M, N, K), 1), device=dev)
        st = torch.cuda.current_stream

        def run():
            for w in W:
                if kc:
                    lib().genrl_sgemm(A.data_ptr(), K, 1, w.data_ptr(), K, 1, C.data_ptr(), N, None, M, N, K, 0, ws.data_ptr(), ws.numel(), st().cuda_stream)
                else:
                    lib().genrl_sgemm(A.data_ptr(), K, 1, w.data_ptr(), 1, N, C.data_ptr(), N, None, M, N, K, 0, ws.data_ptr(), ws.numel(), st().cuda_stream)
        t = min(graph_time(run, n=6) / nset for _ in range(3))
        print(f'{sys.argv[1]:8s} M={M:3d} N={N:5d} K={K:5d} W {"k-contiguous" if kc else "row-contiguous (dgrad)"}: {t:6.2f} us', flush=True)
