cd ${GRAFT_REPO_ROOT:-.}
bash scripts/build_abl.sh 8 9 10 > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -shared -fPIC -DPLANES_NT_STORE -I include -I genrl_amd/csrc -o gpurun_ablnt.so genrl_amd/csrc/*.hip 2>/dev/null
for a in 8 9 10; do GENRL_HIP_SO=$PWD/gpurun_abl$a.so python scripts/intercept64.py $a; done 2>&1 | grep -v amdgpu.ids
GENRL_HIP_SO=$PWD/gpurun_ablnt.so python scripts/intercept64.py nt 2>&1 | grep -v amdgpu.ids
python scripts/intercept64.py 0 2>&1 | grep -v amdgpu.ids | grep -E "x   64|x 1024|x 3072"
