# A/B of two builds of the library on one box: genrl_amd/libgenrl_prev.so (the previous build, copied aside; git-ignored) against the in-tree one
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in ${CONFIGS:-c2 c3 c4 c5}; do for r in 1 2; do
echo "$c: old build: $(GENRL_HIP_SO=$PWD/genrl_amd/libgenrl_prev.so $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   new build: $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
