cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_conv_planes.py tests/test_gpu_planes.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_iteration.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in c2 c4 c3; do for r in 1 2; do
echo "$c: default (128x192 for N=192, 256x96 for N<=96) $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_HL_TALL=0 $(GENRL_HL_TALL=0 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_HL_WIDE=0 GENRL_HL_TALL=0 $(GENRL_HL_WIDE=0 GENRL_HL_TALL=0 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
bash scripts/inshape.sh 2>&1 | grep -E "subpixel|conv128"
