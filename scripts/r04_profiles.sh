#!/bin/bash
# round-4 judged artefacts (written under gpurun_out/r04p/; copy into profiles/ afterwards)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
# (1) clean kernel stats of the default arithmetic, graph replay
rm -rf /tmp/k1; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1 -o p -- $B --steps 6 --warmup 3 > $O/bench_default.json 2> /dev/null
cp /tmp/k1/p_kernel_stats.csv $O/kernel_stats_default.csv 2>/dev/null
python scripts/kernel_table.py /tmp/k1/p_kernel_trace.csv 4 > $O/kernel_table_default.txt 2>&1
python scripts/native_count.py /tmp/k1/p_kernel_trace.csv > $O/launch_classes.txt 2>&1
# (2) feature A/B on this one box
{
echo "default:                          $($B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_SUBPIXEL=0 (GEMM -> col2im): $(GENRL_SUBPIXEL=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_PLANES_CONV=0:              $(GENRL_PLANES_CONV=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "GENRL_PLANES_WGRAD=0:             $(GENRL_PLANES_WGRAD=0 $B --steps 20 --warmup 5 2>/dev/null | ms)"
echo "default again:                    $($B --steps 20 --warmup 5 2>/dev/null | ms)"
} > $O/feature_ab.txt 2>&1
# (3) per-rank batch table, connector side stream ON and OFF
for b in 32 16 8 4; do
  echo "B=$b overlap on: $($B --batch $b --steps 30 2>/dev/null | ms)   no-overlap: $($B --batch $b --steps 30 --no-overlap 2>/dev/null | ms)"
done > $O/batch_table.txt 2>&1
# (4) one bench line per BASELINE config (graph replay + eager leg + per-pipe roofline; no PMC traffic passes except c2 and c4)
for c in c3 c5; do timeout 600 python bench.py --config $c --no-cpu-baseline --no-traffic > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 900 python bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
# (5) per-phase kernels of one eager single-stream step
timeout 300 bash scripts/phase_prof.sh 32 14 > $O/phase_b32.txt 2>&1
timeout 300 bash scripts/phase_prof.sh 4 14 > $O/phase_b4.txt 2>&1
# (6) micro-benchmarks: weight-gradient kernel (XCD-aware order), the 64x64 tile's ablations
timeout 200 python scripts/tn_bench.py > $O/tn_bench.txt 2>&1
# (the 64x64 tile's ablations need the -DPLANES_ABL builds: scripts/build_abl.sh 1 2 3 4, then scripts/abl64.py -- profiles/r04_abl64.txt)
timeout 200 python scripts/fused_small_time.py > $O/fused_small_time.txt 2>&1
timeout 200 python scripts/convt_direct_time.py > $O/convt_direct_time.txt 2>&1
# (7) in-step time of the plane GEMM per shape; PMC passes; the full default bench line (with CPU baseline and traffic)
timeout 300 bash scripts/inshape.sh > $O/inshape.txt 2>&1
timeout 600 bash scripts/pmc.sh > $O/pmc.txt 2>&1; cp gpurun_out/pmc/pmc_summary.json $O/pmc.json 2>/dev/null
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
