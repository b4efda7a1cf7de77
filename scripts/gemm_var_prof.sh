#!/bin/bash
cd $GRAFT_REPO_ROOT
SRC="genrl_amd/csrc/gemm.hip genrl_amd/csrc/rowops.hip genrl_amd/csrc/dist.hip genrl_amd/csrc/conv.hip genrl_amd/csrc/optim.hip"
for v in "64,4,2" "32,2,2" "16,1,2" "32,2,1"; do
  IFS=, read sbk skg spd <<< "$v"
  out=/tmp/lib_v.so; rm -f $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DGENRL_SMALL_BK=$sbk -DGENRL_SMALL_KG=$skg -DGENRL_SMALL_PD=$spd -o $out $SRC 2>&1 | grep -E "error" -A3
  echo "=== small BK=$sbk KG=$skg PD=$spd"
  GENRL_HIP_SO=$out bash scripts/gemm_prof.sh "1024 1024 64 kk" "1024 1024 1024 kk" "1024 1024 2048 kk"
done
