# where the idle gaps of a replayed step are: kernel sequence of its first 2 ms (c2, c4)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in c2; do
rm -rf /tmp/tl; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tl -o p -- python bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --no-eager-leg > /dev/null 2>&1
echo "== $c"; python scripts/timeline.py /tmp/tl/p_kernel_trace.csv 16 2.0
ls /tmp/tl; head -3 /tmp/tl/p_memory_copy_trace.csv 2>/dev/null; wc -l /tmp/tl/p_memory_copy_trace.csv 2>/dev/null
done
