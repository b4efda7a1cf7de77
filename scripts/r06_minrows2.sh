#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
B="timeout 150 python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
{
for r in 1 2; do
echo "c2 B=4 (128 rows): default (fp32 operands below 192 rows) $($B --batch 4 --steps 30 --warmup 5 2>/dev/null | ms)   planes from 128 rows + fused LN $(GENRL_PLANES_MIN_ROWS=128 $B --batch 4 --steps 30 --warmup 5 2>/dev/null | ms)   planes from 128 rows, GENRL_GEMM_LN=0 $(GENRL_PLANES_MIN_ROWS=128 GENRL_GEMM_LN=0 $B --batch 4 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c2 B=2 (64 rows): default $($B --batch 2 --steps 30 --warmup 5 2>/dev/null | ms)   planes from 64 rows + fused LN $(GENRL_PLANES_MIN_ROWS=64 $B --batch 2 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c3 B=8 (400 rows): default $($B --config c3 --batch 8 --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_GEMM_LN=0 $(GENRL_GEMM_LN=0 $B --config c3 --batch 8 --steps 30 --warmup 5 2>/dev/null | ms)"
done
} > $O/minrows2.txt 2>&1
cat $O/minrows2.txt
