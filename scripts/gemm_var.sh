#!/bin/bash
# time GEMM shapes with library variants: scripts/gemm_var.sh "<flags1>" "<flags2>" ... (shapes in $SHAPES)
cd $GRAFT_REPO_ROOT
SRC="genrl_amd/csrc/gemm.hip genrl_amd/csrc/rowops.hip genrl_amd/csrc/dist.hip genrl_amd/csrc/conv.hip genrl_amd/csrc/optim.hip"
SHAPES=${SHAPES:-"1024 1024 1024 kk;1024 1024 1024 kr;1024 1024 1024 rr;1024 3072 1024 kk;4096 1024 1024 kk"}
i=0
for fl in "$@"; do
  i=$((i+1)); out=/tmp/lib_var$i.so; rm -f $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude $fl -o $out $SRC 2>&1 | grep -E "error" -A3
  echo "=== variant [$fl]"
  IFS=';' read -ra SH <<< "$SHAPES"
  for shp in "${SH[@]}"; do GENRL_HIP_SO=$out scripts/gemm_prof.sh "$shp" 2>&1 | grep gemm; done
done
