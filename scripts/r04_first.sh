#!/bin/bash
# round 4, first GPU pass: the -m gpu suite (with the measured parity values reported), then one bench line per BASELINE config
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/parity_measured.txt
GENRL_PARITY_REPORT=$PWD/gpurun_out/r04/parity_measured.txt bash scripts/gpu_suite.sh 2>&1 | tee gpurun_out/r04/suite.txt
for c in c2 c3 c4 c5; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/r04/bench_$c.json 2> gpurun_out/r04/bench_$c.err
  echo "bench $c rc=$? $(head -c 300 gpurun_out/r04/bench_$c.json)"
done
