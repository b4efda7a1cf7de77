#!/bin/bash
# round 5: per-rank batch tables of the configs BASELINE names as multi-GPU (c3, c5), the c5 operand-policy A/B, c3 / c5 lines with PMC traffic
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
{
echo "# bench.py --config c3 (DreamerAgent, dreamer_v3.yaml, T = 50) at the per-rank batch of each data-parallel degree, ONE GPU, hipGraph replay, 30 steps; side streams on / off"
for b in 64 32 16 8; do
  echo "sequences=$b (DP-$((64 / b)) per-rank): overlap on: $($B --config c3 --batch $b --steps 30 2>/dev/null | ms)   no-overlap: $($B --config c3 --batch $b --steps 30 --no-overlap 2>/dev/null | ms)"
done
} > $O/batch_table_c3.txt 2>&1
{
echo "# bench.py --config c5 (data-free block, 256 start rows per GPU = the per-rank size at every DP degree: weak scaling), ONE GPU, hipGraph replay, 30 steps"
echo "256 rows: overlap on: $($B --config c5 --steps 30 2>/dev/null | ms)   no-overlap: $($B --config c5 --steps 30 --no-overlap 2>/dev/null | ms)"
echo "# operand policy (verdict item 6): 256-row rollouts on the fp32-operand kernels (default: GENRL_PLANES_MIN_ROWS=512) against plane operands from 256 rows"
for r in 1 2; do
echo "default (fp32 operands at 256 rows):   $($B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)"
echo "GENRL_PLANES_MIN_ROWS=256 (planes):    $(GENRL_PLANES_MIN_ROWS=256 $B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)"
done
} > $O/batch_table_c5.txt 2>&1
for c in c3 c5; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
cat $O/batch_table_c3.txt $O/batch_table_c5.txt
