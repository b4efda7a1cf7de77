"""Wave-quantisation audit of a rocprofv3 kernel trace: for every kernel (name, grid) the workgroups, the workgroups one CU holds (LDS, registers,
wave slots), the rounds it takes on 256 CUs and the fill of those rounds -- launches that run a mostly empty last round, ranked by the time at stake.
python scripts/round_audit.py <kernel_trace.csv> [top]"""
import csv, sys, collections, re, math
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agg = collections.defaultdict(lambda: [0, 0.0])
info = {}
for r in rows:
    n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[:60]
    wg = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
    grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    nwg = grid // max(wg, 1)
    lds = int(r.get('LDS_Block_Size', 0) or 0)
    vg = int(r.get('VGPR_Count', 0) or 0) + int(r.get('Accum_VGPR_Count', 0) or 0)
    key = (n, nwg)
    agg[key][0] += 1; agg[key][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    info[key] = (wg, lds, vg)
out = []
for (n, nwg), (cnt, us) in agg.items():
    wg, lds, vg = info[(n, nwg)]
    waves = max(wg // 64, 1)
    wps = max(1, min(8, 512 // max(vg, 1))) if vg else 8
    per_cu = min(163840 // lds if lds else 99, (4 * wps) // waves if waves <= 4 * wps else 1, 32 // waves if waves <= 32 else 1)
    per_cu = max(per_cu, 1)
    slots = 256 * per_cu
    rounds = math.ceil(nwg / slots)
    fill = nwg / (rounds * slots)
    if nwg >= 200:
        out.append((us * (1 - fill), n, nwg, per_cu, rounds, fill, cnt, us))
out.sort(reverse=True)
print(f'# {"kernel":60s} {"WGs":>7s} {"/CU":>4s} {"rounds":>6s} {"fill":>5s} {"launches":>8s} {"total us":>10s} {"us at stake":>11s}')
for st, n, nwg, pc, rd, fill, cnt, us in out[:top]:
    print(f'  {n:60s} {nwg:7d} {pc:4d} {rd:6d} {fill:5.2f} {cnt:8d} {us:10.0f} {st:11.0f}')
