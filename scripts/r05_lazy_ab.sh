# inner-layer channel-LayerNorm writing planes only (GENRL_CONV_LAZY_FP32, ops_conv_planes._ln_fwd): parity files, then the step A/B
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_conv_planes.py tests/test_gpu_api.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in c4 c2 c3; do for r in 1 2; do
echo "$c: default $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_CONV_LAZY_FP32=0 $(GENRL_CONV_LAZY_FP32=0 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
