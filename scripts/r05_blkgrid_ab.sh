# workgroups of the row-wise backward kernels (LayerNorm wave / block kernels, GRU gate block): GENRL_BLK_GRID 512 against 768 / 1024
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in c2 c3 c4 c5; do for r in 1 2; do
echo "$c: 512: $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   768: $(GENRL_BLK_GRID=768 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   1024: $(GENRL_BLK_GRID=1024 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
