cd ${GRAFT_REPO_ROOT:-.}
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -shared -fPIC -I include -I genrl_amd/csrc"
$B -DTN_ABL_NOWAIT -o gpurun_abltn0.so genrl_amd/csrc/*.hip 2>/dev/null &
$B -DTN_ABL_NOWAIT -DTN_ABL_DMA=1 -o gpurun_abltn1.so genrl_amd/csrc/*.hip 2>/dev/null &
$B -DTN_ABL_NOWAIT -DTN_ABL_DMA=2 -o gpurun_abltn2.so genrl_amd/csrc/*.hip 2>/dev/null &
$B -DTN_ABL_NOWAIT -DTN_ABL_DMA=3 -o gpurun_abltn3.so genrl_amd/csrc/*.hip 2>/dev/null &
wait
echo "== shipped"; python scripts/tn_conv_probe.py 2>&1 | grep -v amdgpu
echo "== no wait for the DMAs (wrong results)"; GENRL_HIP_SO=$PWD/gpurun_abltn0.so python scripts/tn_conv_probe.py 2>&1 | grep -v amdgpu
echo "== no wait, no tile DMAs in the loop"; GENRL_HIP_SO=$PWD/gpurun_abltn1.so python scripts/tn_conv_probe.py 2>&1 | grep -v amdgpu
echo "== no wait, no B DMAs in the loop"; GENRL_HIP_SO=$PWD/gpurun_abltn2.so python scripts/tn_conv_probe.py 2>&1 | grep -v amdgpu
echo "== no wait, no A DMAs in the loop"; GENRL_HIP_SO=$PWD/gpurun_abltn3.so python scripts/tn_conv_probe.py 2>&1 | grep -v amdgpu
