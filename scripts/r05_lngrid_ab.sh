# workgroups of the channel-LayerNorm backward (GENRL_LN_NARROW_GRID): 512 (2 per CU) against 1024 / 2048 (build with REDUCE2M_MAX raised)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-c++20-extensions -shared -fPIC -I include -I genrl_amd/csrc -DREDUCE2M_MAX=2048 -o gpurun_abl_r2m.so genrl_amd/csrc/*.hip 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_conv_planes.py tests/test_gpu_api.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
export GENRL_HIP_SO=$PWD/gpurun_abl_r2m.so
for c in c4 c2 c3; do for r in 1 2; do
echo "$c: 512: $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   1024: $(GENRL_LN_NARROW_GRID=1024 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   2048: $(GENRL_LN_NARROW_GRID=2048 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done; done
