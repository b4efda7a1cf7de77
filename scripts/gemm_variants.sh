#!/bin/bash
# build GEMM variants on the GPU box and benchmark each (kernel-tuning experiments)
cd $GRAFT_REPO_ROOT
SRC="genrl_amd/csrc/gemm.hip genrl_amd/csrc/rowops.hip genrl_amd/csrc/dist.hip genrl_amd/csrc/conv.hip genrl_amd/csrc/optim.hip"
for v in "$@"; do
  IFS=, read sbk skg bbk bkg spd <<< "$v"
  out=/tmp/lib_${sbk}_${skg}_${bbk}_${bkg}.so; rm -f $out
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -DGENRL_SMALL_BK=$sbk -DGENRL_SMALL_KG=$skg -DGENRL_BIG_BK=$bbk -DGENRL_BIG_KG=$bkg -DGENRL_SMALL_PD=${spd:-2} -o $out $SRC 2>&1 | grep -E "error" -A3
  echo "=== small BK=$sbk KG=$skg PD=${spd:-2}  big BK=$bbk KG=$bkg"
  GENRL_HIP_SO=$out python scripts/gemm_bench.py 2>&1 | grep "TF/s"
done
