"""In-step microseconds of the plane GEMM PER SHAPE: joins a rocprofv3 kernel trace of an eager, single-stream bench run with
the library's own launch log of the same process (GENRL_GEMM_LOG: tile, M, N, K per launch, in launch order -- on one stream the
n-th logged launch IS the n-th gemm_planes_kernel dispatch).  python scripts/inshape_table.py <kernel_trace.csv> <gemm.log>"""
import csv, sys, collections

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_planes_kernel<' in r['Kernel_Name'] or 'gemm_planes_hl_kernel<' in r['Kernel_Name'] or 'gemm_planes_hlw_kernel<' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
log = [tuple(l.split()[:4]) for l in open(sys.argv[2]) if l.strip() and l.startswith(('h2/', 'x3/'))]
assert len(rows) == len(log), (len(rows), len(log))
# one step = the shortest period of the logged sequence at its end
P = next(p for p in range(50, len(log) // 2) if log[-p:] == log[-2 * p:-p])
nsteps = 1
while (nsteps + 1) * P <= len(log) and log[-(nsteps + 1) * P:-nsteps * P] == log[-P:]:
    nsteps += 1
nsteps = min(nsteps, 4)
agg = collections.defaultdict(list)
for r, l in list(zip(rows, log))[-nsteps * P:]:
    wgs = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    agg[l + (wgs,)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values()) / nsteps
print(f'# {P} plane-GEMM launches per step, {tot / 1e3:.2f} ms per step in them (last {nsteps} steps of the trace; eager, one stream)')
print(f'# {"tile":10s} {"M":>7s} {"N":>6s} {"K":>6s} {"workgroups":>10s} {"launches/step":>13s} {"us mean":>8s} {"us min":>8s} {"ms/step":>8s} {"TF/s (2MNK)":>12s}')
for (tile, M, N, K, wgs), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    M, N, K = int(M), int(N), int(K)
    mean = sum(v) / len(v)
    print(f'  {tile:10s} {M:7d} {N:6d} {K:6d} {wgs:10d} {len(v) / nsteps:13.1f} {mean:8.1f} {min(v):8.1f} {sum(v) / nsteps / 1e3:8.3f} {2.0 * M * N * K / mean / 1e6:12.0f}')
