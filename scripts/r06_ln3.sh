#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
B="timeout 120 python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
timeout 100 python scripts/gemm_ln_debug.py 2>&1 | grep "fail word" | cut -c1-60
timeout 300 python -m pytest tests/test_gpu_gemm_ln.py -m gpu -x -q --timeout 100 -p no:cacheprovider > $O/t_ln.log 2>&1; rc=$?; echo "gemm_ln tests rc=$rc"; tail -5 $O/t_ln.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout 200 python scripts/gemm_ln_time.py > $O/gemm_ln_time.txt 2>&1; cat $O/gemm_ln_time.txt
{
for r in 1 2; do
for b in 32 8; do
echo "c2 B=$b: GENRL_GEMM_LN=0 $(GENRL_GEMM_LN=0 $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_GEMM_LN=1 $(GENRL_GEMM_LN=1 $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)"
done
done
echo "c5: GENRL_GEMM_LN=0 $(GENRL_GEMM_LN=0 $B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)   GENRL_GEMM_LN=1 $(GENRL_GEMM_LN=1 $B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)"
} > $O/ln_ab.txt 2>&1
cat $O/ln_ab.txt
timeout 900 python -m pytest tests/test_gpu_iteration.py tests/test_gpu_api.py tests/test_gpu_planes.py -m gpu -q --timeout 200 -p no:cacheprovider > $O/t_paths.log 2>&1; echo "paths rc=$?"; tail -5 $O/t_paths.log
