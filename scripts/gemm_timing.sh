#!/bin/bash
cd $GRAFT_REPO_ROOT
SRC="genrl_amd/csrc/gemm.hip genrl_amd/csrc/rowops.hip genrl_amd/csrc/dist.hip genrl_amd/csrc/conv.hip genrl_amd/csrc/optim.hip"
out=/tmp/lib_t.so; rm -f $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DGENRL_DBG_TIMING "$@" -o $out $SRC 2>&1 | grep -E "error" -A3
for shape in "1024 1024 1024 kk" "1024 1024 1024 rr"; do
  GENRL_HIP_SO=$out python scripts/gemm_timing.py $shape
done
