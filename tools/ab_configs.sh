#!/bin/bash
# GPU box: BASELINE configs 3 / 5 / config-4 shard, this tree against .ab/prev (the round-3 tree), interleaved
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04; mkdir -p $O
for rep in 1 2; do
  for c in ${CASES:-cfg5 cfg3 cfg4_shard}; do
    for t in . .ab/prev; do
      (cd $t && timeout 300 python3 tools/bench_configs.py $c 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-9s' % '$t', json.dumps(d)[:400])
")
    done
  done
done | tee $O/ab_configs.txt
