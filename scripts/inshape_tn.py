"""In-step microseconds of the weight-gradient plane kernel (gemm_planes_tn_kernel) PER SHAPE: the join of scripts/inshape_table.py for the
h2tn / h2tn/conv lines of the launch log.  python scripts/inshape_tn.py <kernel_trace.csv> <gemm.log>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_planes_tn_kernel<' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
log = [tuple(l.split()[:4]) for l in open(sys.argv[2]) if l.startswith('h2tn')]
assert len(rows) == len(log), (len(rows), len(log))
n = len(log)
P = next(p for p in range(4, n // 2 + 1) if log[-p:] == log[-2 * p:-p])
agg = collections.defaultdict(list)
for r, l in list(zip(rows, log))[-2 * P:]:
    wgs = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    agg[l + (wgs,)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f'# {P} weight-gradient launches per step (last 2 steps)')
print(f'# {"family":10s} {"NI":>6s} {"NJ":>6s} {"M":>8s} {"workgroups":>10s} {"launches/step":>13s} {"us mean":>8s} {"ms/step":>8s} {"TF/s (2 NI NJ M)":>16s}')
for (fam, NI, NJ, M, wgs), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    NI, NJ, M = int(NI), int(NJ), int(M)
    mean = sum(v) / len(v)
    print(f'  {fam:10s} {NI:6d} {NJ:6d} {M:8d} {wgs:10d} {len(v) / 2:13.1f} {mean:8.1f} {sum(v) / 2 / 1e3:8.3f} {2.0 * NI * NJ * M / mean / 1e6:16.0f}')
