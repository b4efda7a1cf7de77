cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
for c in c2 c4 c3; do
rm -rf /tmp/ka; rocprofv3 --kernel-trace --output-format csv -d /tmp/ka -o p -- $B --config $c --steps 4 --warmup 2 > /dev/null 2>&1
echo "== $c (whole trace: 7 iterations incl. warm-up)"; python scripts/round_audit.py /tmp/ka/p_kernel_trace.csv 18
done
