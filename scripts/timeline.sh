#!/bin/bash
# scripts/timeline.sh [bench args]: kernel trace of the default (graph, overlapped) bench + scripts/timeline.py
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --no-eager-leg "$@" > /dev/null 2>&1
python scripts/timeline.py /tmp/tl/p_kernel_trace.csv 16
