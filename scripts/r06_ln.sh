#!/bin/bash
# round 6: the fused Dense -> LayerNorm launch: unit tests, the tests of the paths that now take it, A/B in the step at 32 / 16 / 8 sequences
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
timeout 600 python -m pytest tests/test_gpu_gemm_ln.py -m gpu -x -q --timeout 250 -p no:cacheprovider > $O/t_ln.log 2>&1; echo "gemm_ln tests rc=$?"; tail -5 $O/t_ln.log
timeout 1500 python -m pytest tests/test_gpu_planes.py tests/test_gpu_iteration.py tests/test_gpu_api.py tests/test_gpu_conv_planes.py -m gpu -q --timeout 250 -p no:cacheprovider > $O/t_paths.log 2>&1; echo "paths rc=$?"; tail -5 $O/t_paths.log
{
for r in 1 2; do
for b in 32 16 8; do
echo "c2 B=$b: GENRL_GEMM_LN=0 $(GENRL_GEMM_LN=0 $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_GEMM_LN=1 $(GENRL_GEMM_LN=1 $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)"
done
echo "c5: GENRL_GEMM_LN=0 $(GENRL_GEMM_LN=0 $B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)   GENRL_GEMM_LN=1 $(GENRL_GEMM_LN=1 $B --config c5 --steps 50 --warmup 10 2>/dev/null | ms)"
done
} > $O/ln_ab.txt 2>&1
cat $O/ln_ab.txt
