# split-K products of the fp32-operand tile kernel: reduction inside the kernel (GENRL_SPLITK_INKERNEL=1) against partial tiles + reduce launch (default)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in c5 c2 c3 c4; do for r in 1 2; do
echo "$c: in-kernel reduction: $(GENRL_SPLITK_INKERNEL=1 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   GENRL_SPLITK_INKERNEL=0 (reduce launch): $(GENRL_SPLITK_INKERNEL=0 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
for b in 4 8; do
echo "c2 at $b sequences: in-kernel: $(GENRL_SPLITK_INKERNEL=1 $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)   reduce launch: $(GENRL_SPLITK_INKERNEL=0 $B --batch $b --steps 30 --warmup 5 2>/dev/null | ms)"
done
