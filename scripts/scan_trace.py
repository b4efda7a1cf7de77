"""EnsembleRSSM.observe (single_obs_posterior: false, Dreamer-v3 widths) forward + backward alone, for a rocprofv3 kernel trace:
rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o p -- python scripts/scan_trace.py <B> ; python scripts/scan_trace.py --table /tmp/st/p_kernel_trace.csv"""
import sys, os, csv, collections, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == '--table':
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
    # the last of the repeated scans: from the last sgemm batch... keep it simple: aggregate the last third of the trace
    rows = rows[2 * len(rows) // 3:]
    agg = collections.defaultdict(lambda: [0, 0.0])
    gaps = []
    for a, b in zip(rows, rows[1:]):
        gaps.append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
    for r in rows:
        n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
        e = agg[n]; e[0] += 1; e[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot = sum(v[1] for v in agg.values())
    wall = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
    print(f'# {len(rows)} launches, {tot / 1e3:.2f} ms of kernel time in {wall / 1e3:.2f} ms wall; median gap between consecutive kernels {sorted(gaps)[len(gaps) // 2]:.2f} us')
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{n:72s} {c:6d} x {us / c:7.2f} us = {us / 1e3:7.3f} ms')
    sys.exit(0)
import torch
from genrl_amd.agent import dreamer_utils as common
B, T, S, K, D, U, E, A = int(sys.argv[1]), 50, 32, 32, 512, 512, 1536, 6
torch.manual_seed(0)
r = common.EnsembleRSSM(ensemble=1, stoch=S, deter=D, hidden=U, discrete=K, act='SiLU', norm='layer', action_dim=A, embed_dim=E,
                        device='cuda', single_obs_posterior=False).cuda()
embed = torch.randn(B, T, E, device='cuda', requires_grad=True)
action = torch.tanh(torch.randn(B, T, A, device='cuda'))
is_first = torch.zeros(B, T, dtype=torch.bool, device='cuda'); is_first[:, 0] = True
w = torch.randn(B, T, D, device='cuda')


def step():
    for p in r.parameters():
        p.grad = None
    post, prior = r.observe(embed, action, is_first, None)
    ((post['deter'] * w).sum() + post['logit'].sum() * 0.01 + (post['stoch'] * post['logit'].detach()).sum()).backward()


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
print(f'observe fwd + bwd, B={B} T={T}: {e0.elapsed_time(e1) / 10:.3f} ms per replay', flush=True)
