cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python scripts/tn_conv_probe.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_planes.py tests/test_gpu_conv_planes.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in c2 c3 c4; do for r in 1 2; do
echo "$c: default $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_TN_SPLIT_NEAREST=1 $(GENRL_TN_SPLIT_NEAREST=1 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
