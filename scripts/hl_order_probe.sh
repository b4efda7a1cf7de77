cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for xm in auto 8 4 2; do for o in 0 1 2; do
  if [ $xm = auto ]; then unset GENRL_XCD_M; else export GENRL_XCD_M=$xm; fi
  export GENRL_HL_ORDER=$o
  python scripts/hl_order_probe.py 2>&1 | grep -v amdgpu
  rm -rf /tmp/hp; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/hp -o p -- python scripts/hl_order_probe.py --once > /dev/null 2>&1
  python - <<'PY'
import csv, glob
f = glob.glob('/tmp/hp/**/*counter_collection.csv', recursive=True)
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f[0])) if 'gemm_planes_hl_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE'] if f else []
print(f'    read per launch: {sum(v) / max(len(v), 1) * 2 * 1024 / 1e6:.1f} MB over {len(v)} launches')
PY
done; done
