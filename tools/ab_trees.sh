#!/bin/bash
# same-box A/B of two source trees (kernel + host tables): tools/ab_trees.sh <treeA> <treeB> [reps] [bench args]
A=$1; B=$2; R=${3:-3}; shift 3
for rep in $(seq $R); do
  for t in "$A" "$B"; do
    (cd $t && python bench.py --steps 2000 --warmup 200 --repeats 10 --no-cpu-baseline "$@" 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-12s rep $rep: %.1f M env-steps/s  %.3f us/step  events %.3f us/launch  fused %.1f M' % ('$t', d['value'] / 1e6, d['ms_per_step'] * 1e3, d['roofline']['kernel_us'], d.get('fused_rollout', {}).get('env_steps_per_s_per_gpu', 0) / 1e6))")
  done
done
