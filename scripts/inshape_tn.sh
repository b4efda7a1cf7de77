cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in c4 c2; do
GENRL_GEMM_LOG=/tmp/gemm4.log rocprofv3 --kernel-trace --output-format csv -d /tmp/is4 -o p -- python bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --no-eager-leg --graph off --no-overlap > /dev/null 2>&1
echo "== $c"; python scripts/inshape_tn.py /tmp/is4/p_kernel_trace.csv /tmp/gemm4.log
done
