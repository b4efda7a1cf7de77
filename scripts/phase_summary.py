import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# phases end at adam_kernel; take the last complete step (5 adam kernels)
idx = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
last = idx[-6:]          # boundaries of the last 5 phases
names = ['wm', 'conn1', 'conn2', 'actor', 'critic']
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:60]
tot_all = 0
for p in range(5):
    seg = ev[last[p] + 1:last[p + 1] + 1]
    wall = (seg[-1][1] - ev[last[p]][1]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    agg = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n in seg:
        a = agg[short(n)]; a[0] += 1; a[1] += (e - s) / 1e6
    print(f'== {names[p]}: wall {wall:.2f} ms, kernel busy {busy:.2f} ms, {len(seg)} kernels')
    for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
        print(f'     {ms:7.2f} ms {c:5d}x  {n}')
    tot_all += wall
print('step wall', tot_all)
