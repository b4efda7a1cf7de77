# kernel table of the data-free block (c5: 256 start rows, fp32-operand rollout) -- default and with plane operands from 256 rows
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
mkdir -p gpurun_out/r5c
rm -rf /tmp/k5; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o p -- $B --config c5 --steps 6 --warmup 3 > /dev/null 2>&1
python scripts/kernel_table.py /tmp/k5/p_kernel_trace.csv 4 3 > gpurun_out/r5c/kernel_table_c5.txt 2>&1
rm -rf /tmp/k5; GENRL_PLANES_MIN_ROWS=256 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k5 -o p -- $B --config c5 --steps 6 --warmup 3 > /dev/null 2>&1
python scripts/kernel_table.py /tmp/k5/p_kernel_trace.csv 4 3 > gpurun_out/r5c/kernel_table_c5_planes256.txt 2>&1
head -45 gpurun_out/r5c/kernel_table_c5.txt | cut -c1-160; echo; head -40 gpurun_out/r5c/kernel_table_c5_planes256.txt | cut -c1-160
