#!/bin/bash
# the -m gpu suite one FILE per process (an abort -- a GPU fault surfaces as SIGABRT -- then costs one file, not the run);
# logs under gpurun_out/suite/, one summary line per file on stdout
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/suite
for f in tests/test_gpu_*.py; do
  n=$(basename $f .py)
  timeout ${SUITE_TIMEOUT:-600} python -X faulthandler -m pytest $f -m gpu -q --timeout 250 -p no:cacheprovider "$@" > gpurun_out/suite/$n.log 2>&1
  echo "$n rc=$? $(grep -E 'passed|failed|error' gpurun_out/suite/$n.log | tail -1)"
done
