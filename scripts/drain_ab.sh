cd ${GRAFT_REPO_ROOT:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -shared -fPIC -DPLANES_EARLY_DRAIN -I include -I genrl_amd/csrc -o gpurun_abled.so genrl_amd/csrc/*.hip 2>/dev/null
for i in 1 2; do
GENRL_HIP_SO=$PWD/gpurun_abled.so python scripts/drain_ab.py early-drain 2>&1 | grep -v amdgpu.ids
python scripts/drain_ab.py drain-at-end 2>&1 | grep -v amdgpu.ids
done
