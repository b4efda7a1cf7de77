#!/bin/bash
# ablation variants of the plane GEMM (PLANES_ABL: 1 no MFMA, 2 no DMA in the loop, 3 no fragment reads, 4 neither DMA nor reads, 5 no barrier
# either): scripts/build_abl.sh 1 2 3 4 5 && GENRL_HIP_SO=$PWD/gpurun_abl1.so python scripts/cold_bench.py   (results are wrong by construction)
cd "$(dirname "$0")/.."
for a in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -shared -fPIC -DPLANES_ABL=$a -I include -I genrl_amd/csrc -o gpurun_abl$a.so genrl_amd/csrc/*.hip 2>/dev/null & done; wait
ls -la gpurun_abl*.so
