#!/bin/bash
# round-6 judged artefacts (written under gpurun_out/r06p/; copied into profiles/ afterwards).  Every bench call is bounded by `timeout`.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
B="timeout 200 python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
# (1) kernel stats of the default arithmetic, graph replay (c2), + launch classes, + timeline
rm -rf /tmp/k1; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1 -o p -- python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg --steps 6 --warmup 3 > $O/bench_default_under_trace.json 2> /dev/null
cp /tmp/k1/p_kernel_stats.csv $O/kernel_stats_default.csv 2>/dev/null
python scripts/kernel_table.py /tmp/k1/p_kernel_trace.csv 4 > $O/kernel_table_default.txt 2>&1
python scripts/native_count.py /tmp/k1/p_kernel_trace.csv > $O/launch_classes.txt 2>&1
python scripts/timeline.py /tmp/k1/p_kernel_trace.csv 16 > $O/timeline.txt 2>&1
# (2) feature A/B on this one box (this round's switches)
{
echo "default:                                        $($B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_GEMM_LN=0 (Dense and LayerNorm: 2 launches): $(GENRL_GEMM_LN=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "default again:                                  $($B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_GEMM_LN=0 again:                          $(GENRL_GEMM_LN=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "--sync-each-step (host never ahead):            $($B --steps 20 --warmup 5 --sync-each-step 2>/dev/null | ms)"
echo "c3 default:                                     $($B --config c3 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c4 default:                                     $($B --config c4 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c5 default (planes from 192 rows):              $($B --config c5 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c5 GENRL_PLANES_MIN_ROWS=320 (fp32 operands):   $(GENRL_PLANES_MIN_ROWS=320 $B --config c5 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c5 GENRL_GEMM_LN=0:                             $(GENRL_GEMM_LN=0 $B --config c5 --steps 30 --warmup 5 2>/dev/null | ms)"
} > $O/feature_ab.txt 2>&1
# (3) per-rank batch tables, side streams ON and OFF: c2, c3
for b in 32 16 8 4; do
  echo "B=$b overlap on: $($B --batch $b --steps 30 2>/dev/null | ms)   no-overlap: $($B --batch $b --steps 30 --no-overlap 2>/dev/null | ms)   GENRL_GEMM_LN=0: $(GENRL_GEMM_LN=0 $B --batch $b --steps 30 2>/dev/null | ms)"
done > $O/batch_table.txt 2>&1
{
echo "# bench.py --config c3 (DreamerAgent, dreamer_v3.yaml, T = 50) at the per-rank batch of each data-parallel degree, ONE GPU, hipGraph replay, 30 steps"
for b in 64 32 16 8; do
  echo "sequences=$b (DP-$((64 / b)) per-rank): overlap on: $($B --config c3 --batch $b --steps 30 2>/dev/null | ms)"
done
} > $O/batch_table_c3.txt 2>&1
# (4) one bench line per BASELINE config (graph replay + eager leg + per-pipe roofline + PMC traffic with the per-kernel table)
for c in c3 c4 c5; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
# (5) in-step time of the plane GEMM per shape; PMC passes; kernel tables of c3 / c4 / c5; the full default bench line (CPU baseline, traffic)
timeout 300 bash scripts/inshape.sh > $O/inshape.txt 2>&1
timeout 600 bash scripts/pmc.sh > $O/pmc.txt 2>&1; cp gpurun_out/pmc/pmc_summary.json $O/pmc.json 2>/dev/null
for c in c3 c4 c5; do
rm -rf /tmp/kc; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o p -- python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg --config $c --steps 6 --warmup 3 > /dev/null 2>&1
python scripts/kernel_table.py /tmp/kc/p_kernel_trace.csv 4 $([ $c = c3 ] && echo 3) > $O/kernel_table_$c.txt 2>&1
done
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
cat $O/feature_ab.txt; python -c "import json; d=json.load(open('$O/bench_c2.json')); print(d['ms_per_step'], d['config']['eager_ms_per_step'], d['roofline']['dominant_kernel']['name'], d['roofline']['dominant_kernel']['frac'])"
