#!/bin/bash
# A/B on ONE box: scripts/ab.sh "<env A>" "<env B>" [bench args]; prints ms/step alternating, 2 rounds
A="$1"; B="$2"; shift 2
for i in 1 2; do
  for e in "$A" "$B"; do
    v=$(env $e python bench.py --no-cpu-baseline --no-kernel-profile --no-fp32-mode --no-traffic --no-eager-leg --steps 20 "$@" 2>&1 | tail -1 | python -c "import json,sys; print('%.2f ms' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$e] $v"
  done
done
