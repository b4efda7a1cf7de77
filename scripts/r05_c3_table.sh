cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
rm -rf /tmp/k3; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k3 -o p -- $B --config c3 --steps 6 --warmup 3 > gpurun_out/r5c/bench_c3_trace.json 2> /dev/null
python scripts/kernel_table.py /tmp/k3/p_kernel_trace.csv 4 3 > gpurun_out/r5c/kernel_table_c3.txt 2>&1
head -45 gpurun_out/r5c/kernel_table_c3.txt | cut -c1-180
