#!/bin/bash
# which side-stream forks pay where: GENRL_FORK_CRITIC (critic update beside the actor's backward), GENRL_FORK_PRIOR (prior scan beside the decoder)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --no-fp32-mode --no-kernel-profile --no-cpu-baseline --no-traffic --no-eager-leg"
ms() { python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'], 2))"; }
for c in "c2" "c3" "c5" "c4" "c2 --batch 4" "c3 --batch 8"; do
for r in 1 2; do
echo "$c: default $($B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   FORK_CRITIC=0 $(GENRL_FORK_CRITIC=0 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)   FORK_PRIOR=0 $(GENRL_FORK_PRIOR=0 $B --config $c --steps 30 --warmup 5 2>/dev/null | ms)"
done
done
