#!/bin/bash
# round 4: the gather (sub-pixel) form of the transposed convolutions -- parity, then A/B on one box
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
for f in tests/test_gpu_conv_planes.py tests/test_gpu_iteration.py tests/test_gpu_fullsize.py; do
  timeout 600 python -m pytest $f -m gpu -q -x --timeout 250 -p no:cacheprovider 2>&1 | tail -15
done
sed -i 's/--no-traffic --steps 20/--no-traffic --no-eager-leg --steps 20/' scripts/ab.sh
bash scripts/ab.sh "GENRL_SUBPIXEL=0" "GENRL_SUBPIXEL=1" 2>&1 | tee gpurun_out/r04/ab_subpixel.txt
bash scripts/ab.sh "GENRL_SUBPIXEL=1" "GENRL_SUBPIXEL=1 GENRL_SUBPIXEL_ODD=1" 2>&1 | tee -a gpurun_out/r04/ab_subpixel.txt
python bench.py --config c4 --steps 10 --no-cpu-baseline --no-traffic --no-fp32-mode --no-eager-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'])" | tee -a gpurun_out/r04/ab_subpixel.txt
GENRL_SUBPIXEL=0 python bench.py --config c4 --steps 10 --no-cpu-baseline --no-traffic --no-fp32-mode --no-eager-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 subpixel off', d['ms_per_step'])" | tee -a gpurun_out/r04/ab_subpixel.txt
python bench.py --steps 10 --no-cpu-baseline --no-traffic --no-fp32-mode --no-eager-leg --dump-gemm gpurun_out/r04/gemm_shapes.json > gpurun_out/r04/bench_sp.json 2>/dev/null
