#!/bin/bash
# regenerate the judged artefacts under gpurun_out/ (copy into profiles/ afterwards)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 3000 gpurun_out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp /tmp/ks/ks_kernel_stats.csv gpurun_out/bench_kernel_stats.csv
head -12 gpurun_out/bench_kernel_stats.csv | cut -c1-160
