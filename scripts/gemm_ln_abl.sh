#!/bin/bash
# where the fused Dense -> LayerNorm epilogue's time goes: ablation builds of gemm_planes.hip (-DLNE_ABL=n) against the shipped library
cd $GRAFT_REPO_ROOT; python -c "from genrl_amd import build; build.build(force=True, verbose=False)"
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -Wno-unused-value -fPIC -Iinclude -DLNE_ABL=$n -c genrl_amd/csrc/gemm_planes.hip -o /tmp/gp_abl$n.o 2>&1 | grep -E "error" -A3
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl_ln$n.so /tmp/gp_abl$n.o $(ls genrl_amd/csrc/build/*.o | grep -v gemm_planes.o) &
done; wait
echo "shipped:"; timeout 120 python scripts/gemm_ln_time.py 2>&1 | grep -E "^ *(1024|128) x 1024"
for n in 1 2 3 4; do echo "LNE_ABL=$n ($(sed -n "s/.*ablations of the LayerNorm epilogue.*: \(.*\) \*\/.*/\1/p" genrl_amd/csrc/gemm_planes.hip)):"; GENRL_HIP_SO=$PWD/gpurun_abl_ln$n.so timeout 120 python scripts/gemm_ln_time.py 2>&1 | grep -E "^ *(1024|128) x 1024"; done
