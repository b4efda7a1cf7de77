# weight planes / permuted conv weights rebuilt on a side stream at the iteration's start (planes.prefetch; GENRL_WPREFETCH=0: at first use)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in c2 c3 c4 c5; do for r in 1 2; do
echo "$c: GENRL_WPREFETCH=1 (weight planes + permuted conv weights): $(GENRL_WPREFETCH=1 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   =2 (weight planes only): $(GENRL_WPREFETCH=2 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   off (default): $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
