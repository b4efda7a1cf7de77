#!/bin/bash
# final pass at HEAD: whole -m gpu suite in ONE process (as the driver runs it), smoke, the full default bench line
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 300 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench rc=$?"
python -c "import json; d=json.load(open('$O/bench_c2.json')); dk=d['roofline']['dominant_kernel']; print(d['ms_per_step'], d['config']['eager_ms_per_step'], dk['name'], dk['launches'], round(dk['ms_per_step'],2), round(dk['frac'],3), dk['of_which_64x64_launches_with_the_LayerNorm_epilogue']['launches'])"
