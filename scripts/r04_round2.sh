#!/bin/bash
# round 4, second pass: sub-pixel + weight caches parity; TN threshold A/B; tn kernel timing + FETCH_SIZE before/after the XCD-aware order
cd /tmp; export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
for f in tests/test_gpu_conv_planes.py tests/test_gpu_iteration.py tests/test_gpu_planes.py tests/test_gpu_api.py tests/test_gpu_multistep.py; do
  timeout 600 python -m pytest $f -m gpu -q -x --timeout 250 -p no:cacheprovider 2>&1 | tail -3
done
python scripts/tn_bench.py 16384x1024x1024 1024x1024x1024 1024x3072x1024 2>&1 | tee gpurun_out/r04/tn_bench.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/tnp -o p -- python scripts/tn_bench.py 16384x1024x1024 > /dev/null 2>&1
python - <<'PY' | tee -a gpurun_out/r04/tn_bench.txt
import csv, collections
c = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open('/tmp/tnp/p_counter_collection.csv')):
    if r['Counter_Name'] == 'FETCH_SIZE':
        k = r['Kernel_Name'].split('(')[0][-60:]
        c[k] += float(r['Counter_Value']); n[k] += 1
for k in c:
    print(f'FETCH_SIZE x2 per launch: {k:60s} n={n[k]:4d} {2 * 1024 * c[k] / n[k] / 1e6:9.1f} MB')
PY
bash scripts/ab.sh "GENRL_TN_MIN_ROWS=2048" "GENRL_TN_MIN_ROWS=1024" 2>&1 | tee gpurun_out/r04/ab_tn_rows.txt
