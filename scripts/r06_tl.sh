#!/bin/bash
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
rm -rf /tmp/tl; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --no-eager-leg > /dev/null 2>&1
python scripts/timeline.py /tmp/tl/p_kernel_trace.csv 16 2.0 > $O/timeline_head.txt 2>&1
head -30 $O/timeline_head.txt
# the same eager, one stream, no graph: are the holes there too?
rm -rf /tmp/tl2; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl2 -o p -- python bench.py --graph off --no-overlap --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --no-eager-leg > /dev/null 2>&1
python scripts/timeline.py /tmp/tl2/p_kernel_trace.csv 4 2.0 > $O/timeline_head_eager.txt 2>&1
head -14 $O/timeline_head_eager.txt
