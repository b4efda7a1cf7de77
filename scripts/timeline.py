"""Multi-stream timeline of the last training step in a rocprofv3 kernel trace (graph replay, overlapped streams):
wall time, time with no kernel running, the largest idle gaps, and the kernels that run alone (critical path)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')) for r in rows)
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:56]
# a step = from one gather_windows (replay sample, first kernel of a step) to the next
starts = [i for i, e in enumerate(ev) if 'gather_windows' in e[2]]
marks = [starts[0]] + [s for p, s in zip(starts, starts[1:]) if ev[s][0] - ev[p][0] > 5e6]
a, b = marks[-2], marks[-1]
seg = ev[a:b]
t0, t1 = seg[0][0], max(e for _, e, _, _ in seg)
print(f'step wall {(ev[b][0] - t0) / 1e6:.2f} ms, {len(seg)} kernels, busy sum {sum(e - s for s, e, _, _ in seg) / 1e6:.2f} ms')
# sweep: concurrency profile
pts = sorted([(s, 1, n) for s, e, n, _ in seg] + [(e, -1, n) for s, e, n, _ in seg])
conc = collections.Counter(); alone = collections.Counter(); cur = 0; last = t0; running = []
gaps = []
active = {}
for t, d, n in pts:
    conc[min(cur, 4)] += t - last
    if cur == 1 and active:
        alone[short(next(iter(active)))] += t - last
    if cur == 0 and t - last > 0:
        gaps.append((t - last, last, n))
    last = t
    if d == 1:
        active[n] = active.get(n, 0) + 1
    else:
        active[n] -= 1
        if active[n] == 0: del active[n]
    cur += d
for k in sorted(conc): print(f'  {k} kernels running: {conc[k] / 1e6:7.2f} ms')
print('largest idle gaps (us, next kernel):')
for g, at, n in sorted(gaps, reverse=True)[:8]: print(f'   {g / 1e3:7.1f} us at +{(at - t0) / 1e6:6.2f} ms -> {short(n)}')
print(f'idle total {sum(g for g, _, _ in gaps) / 1e6:.2f} ms in {len(gaps)} gaps')
print('time running alone, by kernel:')
for n, t in alone.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14): print(f'   {t / 1e6:7.2f} ms  {n}')
if len(sys.argv) > 3:          # argv[3] = ms: the kernel sequence of the step's first ms (start, duration, gap before, queue)
    lim = float(sys.argv[3]) * 1e6
    prev_end = t0
    print(f'first {sys.argv[3]} ms of the step:')
    for s, e, n, q in seg:
        if s - t0 > lim: break
        print(f'  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}  q{q}  {short(n)}')
        prev_end = max(prev_end, e)
