#!/bin/bash
# calibration sweep of (tile config, split count) per shape: FORCES="s,1 b,4" scripts/gemm_force.sh "M N K mode" ...
cd $GRAFT_REPO_ROOT
FORCES=${FORCES:-"s,1 s,4 s,9 s,14 s,19 s,32 s,64 b,1 b,6 b,18 b,27 b,54 b,64 b,128"}
for shape in "$@"; do
  python scripts/gemm_timeit.py $shape
  for f in $FORCES; do
    GENRL_GEMM_FORCE=$f python scripts/gemm_timeit.py $shape 2>&1 | tail -1
  done
done
