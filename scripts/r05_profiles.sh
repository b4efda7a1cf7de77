#!/bin/bash
# round-5 judged artefacts (written under gpurun_out/r05p/; copied into profiles/ afterwards)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
# (1) kernel stats of the default arithmetic, graph replay (c2), + launch classes
rm -rf /tmp/k1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1 -o p -- $B --steps 6 --warmup 3 > $O/bench_default_under_trace.json 2> /dev/null
cp /tmp/k1/p_kernel_stats.csv $O/kernel_stats_default.csv 2>/dev/null
python scripts/kernel_table.py /tmp/k1/p_kernel_trace.csv 4 > $O/kernel_table_default.txt 2>&1
python scripts/native_count.py /tmp/k1/p_kernel_trace.csv > $O/launch_classes.txt 2>&1
# (2) feature A/B on this one box (this round's switches)
{
echo "default:                                   $($B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_HL_ORDER=0 (row-major XCD sub-block): $(GENRL_HL_ORDER=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_PLANES_2PER=0:                       $(GENRL_PLANES_2PER=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_FORK_CRITIC=0:                       $(GENRL_FORK_CRITIC=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "default again:                             $($B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c3 default:                                $($B --config c3 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c3 GENRL_OBSERVE_SEQ=0 (stepwise observe):  $(GENRL_OBSERVE_SEQ=0 $B --config c3 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c3 GENRL_OBSERVE_FUSE=0 (8 launches/step):  $(GENRL_OBSERVE_FUSE=0 $B --config c3 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c3 GENRL_PLANES_2PER=0:                    $(GENRL_PLANES_2PER=0 $B --config c3 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c3 GENRL_FORK_CRITIC=1:                    $(GENRL_FORK_CRITIC=1 $B --config c3 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c4 default:                                $($B --config c4 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c4 GENRL_TN_SPLIT_NEAREST=1 (round-4 split): $(GENRL_TN_SPLIT_NEAREST=1 $B --config c4 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c4 GENRL_SUBPIXEL_ODD=0:                   $(GENRL_SUBPIXEL_ODD=0 $B --config c4 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c4 GENRL_CONV_LAZY_FP32=0:                 $(GENRL_CONV_LAZY_FP32=0 $B --config c4 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c4 GENRL_LN_NARROW_GRID=512:               $(GENRL_LN_NARROW_GRID=512 $B --config c4 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c4 default again:                          $($B --config c4 --steps 20 --warmup 5 2>/dev/null | ms)"
echo "c5 default:                                $($B --config c5 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c5 GENRL_OBSERVE_SEQ=0 (stepwise imagine):  $(GENRL_OBSERVE_SEQ=0 $B --config c5 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c5 GENRL_FORK_CRITIC=1:                    $(GENRL_FORK_CRITIC=1 $B --config c5 --steps 30 --warmup 5 2>/dev/null | ms)"
} > $O/feature_ab.txt 2>&1
# (3) per-rank batch tables, side streams ON and OFF: c2 and c3
for b in 32 16 8 4; do
  echo "B=$b overlap on: $($B --batch $b --steps 30 2>/dev/null | ms)   no-overlap: $($B --batch $b --steps 30 --no-overlap 2>/dev/null | ms)"
done > $O/batch_table.txt 2>&1
{
echo "# bench.py --config c3 (DreamerAgent, dreamer_v3.yaml, T = 50) at the per-rank batch of each data-parallel degree, ONE GPU, hipGraph replay, 30 steps"
for b in 64 32 16 8; do
  echo "sequences=$b (DP-$((64 / b)) per-rank): overlap on: $($B --config c3 --batch $b --steps 30 2>/dev/null | ms)   no-overlap: $($B --config c3 --batch $b --steps 30 --no-overlap 2>/dev/null | ms)"
done
} > $O/batch_table_c3.txt 2>&1
{
echo "# bench.py --config c5 (data-free block, 256 start rows per GPU = the per-rank size at every DP degree: weak scaling), ONE GPU, hipGraph replay, 30 steps"
echo "256 rows: overlap on: $($B --config c5 --steps 30 2>/dev/null | ms)   no-overlap: $($B --config c5 --steps 30 --no-overlap 2>/dev/null | ms)"
echo "# operand policy (round-4 verdict item 6): 256-row rollouts on the fp32-operand kernels (default: plane operands from 320 rows) against plane operands from 256 rows"
for r in 1 2; do
echo "default (fp32 operands at 256 rows):   $($B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)"
echo "GENRL_PLANES_MIN_ROWS=256 (planes):    $(GENRL_PLANES_MIN_ROWS=256 $B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)"
done
} > $O/batch_table_c5.txt 2>&1
# (4) one bench line per BASELINE config (graph replay + eager leg + per-pipe roofline + PMC traffic with the per-kernel table)
for c in c3 c4 c5; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
# (5) per-phase kernels of one eager single-stream step
timeout 300 bash scripts/phase_prof.sh 32 14 > $O/phase_b32.txt 2>&1
timeout 300 bash scripts/phase_prof.sh 4 14 > $O/phase_b4.txt 2>&1
# (6) in-step time of the plane GEMM per shape; PMC passes; kernel table of c3; the full default bench line (CPU baseline, traffic)
timeout 300 bash scripts/inshape.sh > $O/inshape.txt 2>&1
timeout 600 bash scripts/pmc.sh > $O/pmc.txt 2>&1; cp gpurun_out/pmc/pmc_summary.json $O/pmc.json 2>/dev/null
rm -rf /tmp/k3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k3 -o p -- $B --config c3 --steps 6 --warmup 3 > /dev/null 2>&1
python scripts/kernel_table.py /tmp/k3/p_kernel_trace.csv 4 3 > $O/kernel_table_c3.txt 2>&1
rm -rf /tmp/k4; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -o p -- $B --config c4 --steps 6 --warmup 3 > /dev/null 2>&1
python scripts/kernel_table.py /tmp/k4/p_kernel_trace.csv 4 > $O/kernel_table_c4.txt 2>&1
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err
tail -3 $O/feature_ab.txt; python -c "import json; d=json.load(open('$O/bench_c2.json')); print(d['ms_per_step'], d['config']['eager_ms_per_step'])"
