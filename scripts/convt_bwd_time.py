"""graph-timed input-gradient and weight-gradient kernels of the decoder's 3-channel last layer, each alone (genrl_convt_small_co_bwd with
dWp = NULL / dx = NULL): python scripts/convt_bwd_time.py [images]   (GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genrl_amd._lib import lib
from small_m import graph_time
N, Hi, Ci, Co, k = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 30, 48, 3, 6
dev = 'cuda'
x = torch.randn(N, Hi, Hi, Ci, device=dev); Wp = torch.randn(Ci, k * k * Co, device=dev) * 0.05
dy = torch.randn(N, Co, 64, 64, device=dev); dx = torch.empty_like(x); dW = torch.empty_like(Wp)
ws = torch.empty(lib().genrl_convt_small_co_bwd_ws_floats(Ci, Co), device=dev)
st = torch.cuda.current_stream
f_d = lambda: lib().genrl_convt_small_co_bwd(x.data_ptr(), Wp.data_ptr(), dy.data_ptr(), dx.data_ptr(), None, None, N, Hi, Hi, Ci, Co, k, st().cuda_stream)
f_w = lambda: lib().genrl_convt_small_co_bwd(x.data_ptr(), Wp.data_ptr(), dy.data_ptr(), None, dW.data_ptr(), ws.data_ptr(), N, Hi, Hi, Ci, Co, k, st().cuda_stream)
print(f'{N} images: input gradient {graph_time(f_d, n=10, reps=10):.1f} us   weight gradient (+ reduce) {graph_time(f_w, n=10, reps=10):.1f} us')
