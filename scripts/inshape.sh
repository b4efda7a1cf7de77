#!/bin/bash
# in-step time of the plane GEMM per shape (scripts/inshape_table.py) next to the same shapes back to back / rotating over cold operands
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GENRL_GEMM_LOG=/tmp/gemm.log rocprofv3 --kernel-trace --output-format csv -d /tmp/is -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --graph off --no-overlap > /dev/null 2>&1
python scripts/inshape_table.py /tmp/is/p_kernel_trace.csv /tmp/gemm.log
