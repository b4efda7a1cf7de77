"""Would the imagination rollout gain from running as TWO (or four) independent row blocks on separate streams?  A chain of 32 dependent
(plane GEMM 1024-wide + LayerNorm/SiLU emitting planes) pairs -- the rollout's policy layers -- at N rows on one stream against the same rows
split into 2 / 4 blocks, each block's chain on its own stream, captured as one hipGraph (parallel branches).  Graph-timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd import planes, ops_planes
from genrl_amd._lib import lib

torch.manual_seed(0)
U, L = 1024, 32
Ws = [torch.nn.Parameter(torch.randn(U, U, device='cuda') / 32) for _ in range(4)]
bs = [torch.zeros(U, device='cuda') for _ in range(4)]
g, be = torch.ones(U, device='cuda'), torch.zeros(U, device='cuda')
for w in Ws:
    planes.weight(w)          # split once


def chain(x):
    with torch.no_grad():
        for l in range(L):
            x = ops_planes.dense_ln_act(x, None, Ws[l % 4], bs[l % 4], g, be)
    return x


def timed(N, nblk):
    xs = []
    for i in range(nblk):
        x = torch.randn(N // nblk, U, device='cuda')
        x._planes = (planes.split(x), 0)
        xs.append(x)
    streams = [torch.cuda.Stream() for _ in range(nblk)]
    main = torch.cuda.Stream()

    def run():
        cur = torch.cuda.current_stream()
        for s, x in zip(streams[1:], xs[1:]):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                chain(x)
        chain(xs[0])
        for s in streams[1:]:
            cur.wait_stream(s)
    with torch.cuda.stream(main):
        run(); run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=main):
            run()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(10):
                gr.replay()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
    return best


os.environ['GENRL_PLANES_MIN_ROWS'] = '0'
for N in (1024, 3200):
    for nblk in (1, 2, 4):
        if (N // nblk) % 64:
            continue
        t = timed(N, nblk)
        print(f'{N} rows as {nblk} block(s) of {N // nblk}: {t * 1e3 / L:7.2f} us per (GEMM + LayerNorm) pair, {t:.3f} ms per chain of {L}', flush=True)
