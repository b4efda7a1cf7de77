# the 3-channel transposed-convolution forward kernel taken apart (ablation builds; results of 1-3 are wrong by construction)
cd ${GRAFT_REPO_ROOT:-.}
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -Wno-unused-value -shared -fPIC -I include -I genrl_amd/csrc"
for a in 1 2 3; do $B -DCONVT_ABL=$a -o gpurun_ablc$a.so genrl_amd/csrc/*.hip 2>/dev/null & done; wait
for n in 800 4096; do
echo "== N=$n shipped";            python scripts/convt_direct_time.py $n 1 2>&1 | grep direct
echo "== N=$n no MFMAs";           GENRL_HIP_SO=$PWD/gpurun_ablc1.so python scripts/convt_direct_time.py $n 1 2>&1 | grep direct
echo "== N=$n no loads in the loop"; GENRL_HIP_SO=$PWD/gpurun_ablc2.so python scripts/convt_direct_time.py $n 1 2>&1 | grep direct
echo "== N=$n no epilogue";        GENRL_HIP_SO=$PWD/gpurun_ablc3.so python scripts/convt_direct_time.py $n 1 2>&1 | grep direct
done
# the backward kernels (the input-gradient kernel's ablations 11-13 belong to its round-4 form: profiles/r05_convt_abl.txt)
for a in 21 22 23; do $B -DCONVT_ABL=$a -o gpurun_ablc$a.so genrl_amd/csrc/*.hip 2>/dev/null & done; wait
echo "== backward, shipped";                        python scripts/convt_bwd_time.py 4096 2>&1 | grep images
echo "== wgrad without MFMAs";                      GENRL_HIP_SO=$PWD/gpurun_ablc21.so python scripts/convt_bwd_time.py 4096 2>&1 | grep images
echo "== wgrad without the dy-patch loads";         GENRL_HIP_SO=$PWD/gpurun_ablc22.so python scripts/convt_bwd_time.py 4096 2>&1 | grep images
echo "== wgrad without the x loads";                GENRL_HIP_SO=$PWD/gpurun_ablc23.so python scripts/convt_bwd_time.py 4096 2>&1 | grep images
