#!/bin/bash
# the library with the plane GEMM's experimental variants compiled in (ring depth 2 / 4 / 5, L2 prefetch distances; -DPLANES_EXPERIMENTS):
#   scripts/build_exp.sh && GENRL_HIP_SO=gpurun_exp.so python scripts/cold_bench.py
cd "$(dirname "$0")/.."
SRC=$(ls genrl_amd/csrc/*.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -shared -fPIC -DPLANES_EXPERIMENTS -I include -I genrl_amd/csrc -o gpurun_exp.so $SRC 2>&1 | grep -E " error" -A3
ls -la gpurun_exp.so
