cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/tp; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tp -o p -- python scripts/tn_conv_probe.py 5 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/tp/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'gemm_planes_tn_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
rows.sort(key=lambda r: int(r['Dispatch_Id']))
agg = collections.OrderedDict()
for r in rows:
    key = (r['Kernel_Name'].split('gemm_planes_tn_kernel')[1][:7], r['Grid_Size'])
    agg.setdefault(key, []).append(float(r['Counter_Value']) * 2 * 1024 / 1e6)
for k, v in agg.items():
    print(k, f'{len(v)} launches, read {sum(v) / len(v):8.1f} MB per launch')
PY
