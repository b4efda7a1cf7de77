cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for r in 1 2; do
echo "c3 B=8 (400 rows): default $($B --config c3 --batch 8 --steps 30 --warmup 5 2>/dev/null | ms)  MIN_ROWS=320 $(GENRL_PLANES_MIN_ROWS=320 $B --config c3 --batch 8 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c2 B=8 (256 rows): default $($B --config c2 --batch 8 --steps 30 --warmup 5 2>/dev/null | ms)  MIN_ROWS=256 $(GENRL_PLANES_MIN_ROWS=256 $B --config c2 --batch 8 --steps 30 --warmup 5 2>/dev/null | ms)"
echo "c2 B=12 (384 rows): default $($B --config c2 --batch 12 --steps 30 --warmup 5 2>/dev/null | ms)  MIN_ROWS=320 $(GENRL_PLANES_MIN_ROWS=320 $B --config c2 --batch 12 --steps 30 --warmup 5 2>/dev/null | ms)"
done
