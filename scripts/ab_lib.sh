#!/bin/bash
# A/B two builds of the library on ONE box: scripts/ab_lib.sh "<-D flags for variant B>" [bench args]
cd $GRAFT_REPO_ROOT
SRC="genrl_amd/csrc/gemm.hip genrl_amd/csrc/rowops.hip genrl_amd/csrc/dist.hip genrl_amd/csrc/conv.hip genrl_amd/csrc/optim.hip"
FLAGS="$1"; shift
out=/tmp/lib_b.so; rm -f $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $FLAGS -o $out $SRC 2>&1 | grep -E "error" -A3
scripts/ab.sh "X=default" "GENRL_HIP_SO=$out" "$@"
