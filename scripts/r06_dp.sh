#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
for i in 1 2 3; do
  timeout 700 python -m pytest tests/test_gpu_zz_rccl.py -m gpu -x -q --timeout 650 -p no:cacheprovider > $O/rccl_$i.log 2>&1; echo "rccl run $i rc=$? $(tail -1 $O/rccl_$i.log)"
done
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_dp.py -m gpu -x -q --timeout 500 -p no:cacheprovider > $O/dp.log 2>&1; echo "dp rc=$? $(tail -1 $O/dp.log)"
