cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
rm -rf /tmp/k4; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k4 -o p -- $B --config c4 --steps 6 --warmup 3 > /dev/null 2>&1
python scripts/kernel_table.py /tmp/k4/p_kernel_trace.csv 4 > gpurun_out/r5c/kernel_table_c4.txt 2>&1
head -40 gpurun_out/r5c/kernel_table_c4.txt | cut -c1-180
GENRL_GEMM_LOG=/tmp/gemm4.log rocprofv3 --kernel-trace --output-format csv -d /tmp/is4 -o p -- python bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --no-eager-leg --graph off --no-overlap > /dev/null 2>&1
python scripts/inshape_table.py /tmp/is4/p_kernel_trace.csv /tmp/gemm4.log | head -40
