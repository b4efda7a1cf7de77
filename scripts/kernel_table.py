"""rocprofv3 kernel trace of a bench run -> per-STEP table: for every kernel name the launches per step, the microseconds per
launch and the milliseconds per step, over the LAST n replayed steps of the trace (warm-up, capture and any other run-in are cut:
steps are delimited by the adam_kernel launches, 5 per iteration at c2 / c4, 3 for the DreamerAgent of c3, 2 at c5).
python scripts/kernel_table.py <kernel_trace.csv> [steps [adam launches per iteration]]"""
import csv, sys, collections, re

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
API = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
assert len(adam) >= API * (nsteps + 1), len(adam)
lo, hi = adam[-API * nsteps - 1] + 1, adam[-1] + 1           # kernels of the last nsteps iterations (ordered by start time)
seg = ev[lo:hi]
wall = (seg[-1][1] - ev[lo - 1][1]) / 1e6 / nsteps


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*$', '', n)
    return n[:96]


agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in seg:
    a = agg[short(n)]; a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f'# last {nsteps} steps of the trace: {wall:.2f} ms wall per step (first kernel start to last kernel end, concurrent streams), '
      f'{tot / nsteps / 1e3:.2f} ms of kernel time per step, {sum(v[0] for v in agg.values()) / nsteps:.0f} launches per step')
print(f'# {"kernel":96s} {"launches/step":>13s} {"us/launch":>10s} {"ms/step":>8s} {"% of kernel time":>8s}')
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{n:98s} {c / nsteps:13.1f} {us / c:10.1f} {us / nsteps / 1e3:8.3f} {100 * us / tot:8.2f}')
