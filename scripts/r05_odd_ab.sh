cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in c4 c2 c3; do for r in 1 2; do
echo "$c: default $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_SUBPIXEL_ODD=1 $(GENRL_SUBPIXEL_ODD=1 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
