"""Would two half-batch rollouts on two streams beat one full-batch chain?  Chain of dependent GEMM+LN pairs:
(a) M=1024 on one stream, (b) 2 x M=512 on two streams (graph-captured, like the product)."""
import sys; sys.path.insert(0, '.')
import torch
from genrl_amd import ops
dev = 'cuda'
def chain(x, Ws, g, b, n):
    for i in range(n):
        x = ops.dense_ln_act(x, None, Ws[i % len(Ws)], None if False else bias, g, b)
    return x
N, D, L = 1024, 1024, 40
Ws = [torch.randn(D, D, device=dev) / 32 for _ in range(4)]
bias = torch.zeros(D, device=dev); g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
x = torch.randn(N, D, device=dev)
def run_full():
    with torch.no_grad():
        return chain(x, Ws, g, b, L)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run_dual():
    with torch.no_grad():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = chain(x[:N // 2], Ws, g, b, L)
        with torch.cuda.stream(s2):
            c = chain(x[N // 2:], Ws, g, b, L)
        cur.wait_stream(s1); cur.wait_stream(s2)
        return a, c
def graphed(fn):
    fn(); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        out = fn()
    return gph
def t(gph, reps=20):
    for _ in range(3): gph.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g1 = graphed(run_full); g2 = graphed(run_dual)
print(f'full-batch chain  ({L} x [1024x1024x1024 GEMM + LN]): {t(g1):.3f} ms')
print(f'dual half-batches ({L} x 2 x [512x1024x1024 GEMM + LN]): {t(g2):.3f} ms')
