#!/bin/bash
# profiles/r05_intercept64.txt: scripts/intercept64.py on the shipped library and on the ablation builds 6, 7, 8
cd ${GRAFT_REPO_ROOT:-.}
bash scripts/build_abl.sh 6 7 8 > /dev/null 2>&1
mkdir -p gpurun_out
{
  echo "# scripts/intercept64.py (round 5, verdict item 3): the 64x64 plane tile's fixed cost per launch; graph-timed (50 launches per graph), one MI355X box"
  python scripts/intercept64.py 0
  for a in 6 7 8; do GENRL_HIP_SO=$PWD/gpurun_abl$a.so python scripts/intercept64.py $a; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_intercept64.txt
