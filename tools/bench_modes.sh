#!/bin/bash
# GPU box: the headline line under the three ways of issuing a K-step region (overlapped launches, graph replay, direct launches)
# at the driver's K = 20 and at K = 2000, plus the default line with every leg
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1: value %.1f M  ms_per_step %.4f  kernel_us %.3f  api %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_us'], d['config']['api'][:120]))
"; }
for K in 20 2000; do
  W=$((K/4))
  for rep in 1 2; do
    CZ_CHAIN=1 timeout 300 python3 bench.py --steps $K --warmup $W --no-extras --no-cpu-baseline 2>/dev/null | show "K=$K overlapped rep $rep"
    CZ_CHAIN=0 timeout 300 python3 bench.py --steps $K --warmup $W --no-extras --no-cpu-baseline 2>/dev/null | show "K=$K graph rep $rep"
    CZ_CHAIN=0 CZ_GRAPHS=0 timeout 300 python3 bench.py --steps $K --warmup $W --no-extras --no-cpu-baseline 2>/dev/null | show "K=$K direct rep $rep"
  done
done | tee $O/bench_modes.txt
timeout 900 python3 bench.py --no-cpu-baseline > $O/bench_full_a.json 2>$O/bench_full_a.err; python3 -c "
import json
d=json.load(open('$O/bench_full_a.json'))
for k in ('fused_rollout','fused_actions','closed_loop','closed_loop_compact','cooking_policy'):
    v=d.get(k,{}); print(k, {kk: v[kk] for kk in v if kk in ('ms_per_step','us_per_step','us_per_launch','env_steps_per_s','env_steps_per_s_per_gpu','error')}, v.get('roofline',{}).get('frac'))
print('value', d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac'])
"
