"""Kernel sequence of the last step in a rocprofv3 kernel trace: name, duration, gap to the previous kernel (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    return n.split('(')[0][:70]
starts = [i for i, e in enumerate(ev) if 'gather_windows' in e[2]]
marks = [starts[0]] + [s for p, s in zip(starts, starts[1:]) if ev[s][0] - ev[p][0] > 5e6]
a, b = marks[-2], marks[-1]
prev = ev[a][0]
for s, e, n in ev[a:b]:
    print(f'{(s - ev[a][0]) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us gap {(s - prev) / 1e3:6.1f}  {short(n)}')
    prev = e
