#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
B="timeout 150 python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
{
for r in 1 2; do
echo "c2 default (host enqueues ahead): $($B --steps 30 --warmup 5 2>/dev/null | ms)   --sync-each-step: $($B --steps 30 --warmup 5 --sync-each-step 2>/dev/null | ms)"
done
} > $O/sync_ab.txt 2>&1; cat $O/sync_ab.txt
SUITE_TIMEOUT=500 bash scripts/gpu_suite.sh 2>&1 | tee $O/suite_summary.txt
