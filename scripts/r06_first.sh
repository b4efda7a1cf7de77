#!/bin/bash
# round 6, first GPU call: the changed tests, the bench line with the new roofline fields, the operand-policy A/B at 128 / 192 / 256 rows, CPU thread legs
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_planes.py tests/test_gpu_conv_planes.py -m gpu -x -q --timeout 250 -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python bench.py --no-cpu-baseline --no-traffic > $O/bench_c2.json 2> $O/bench_c2.err; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['ms_per_step'], d['config']['eager_ms_per_step'], d['step_roofline'], d['roofline']['dominant_kernel'], d['roofline']['all_gemm'])"
{
for r in 1 2; do
for b in 4 6 8; do
echo "c2 B=$b ($((b*32)) rows): MIN_ROWS=320 (fp32 operands) $(GENRL_PLANES_MIN_ROWS=320 $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)  MIN_ROWS=$((b*32)) (planes) $(GENRL_PLANES_MIN_ROWS=$((b*32)) $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)"
done
echo "c5 (256 rows): MIN_ROWS=320 $(GENRL_PLANES_MIN_ROWS=320 $B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)  default (256: planes) $($B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)"
done
} > $O/minrows_ab.txt 2>&1
cat $O/minrows_ab.txt
timeout 400 python scripts/cpu_threads.py > $O/cpu_threads.txt 2>&1; cat $O/cpu_threads.txt
