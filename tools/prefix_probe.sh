#!/bin/bash
# GPU box: what does the host latency of a graph replay cost a short (K = 20) timed region, and does launching the first
# few steps directly in front of the graph (CZ_RING_PREFIX) hide it?  usage: bash tools/prefix_probe.sh
for p in 0 1 2 3 4 6; do
  CZ_RING_PREFIX=$p python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('prefix $p: %.1f M env-steps/s  wall %.3f us/step (min %.3f)  events %.3f us/launch  | %s' % (d['value'] / 1e6, d['ms_per_step'] * 1e3, d['ms_per_step_min'] * 1e3, d['roofline']['kernel_us'], d['config']['api'][52:130]))"
done
CZ_GRAPHS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('graphs off: %.1f M env-steps/s  wall %.3f us/step (min %.3f)  events %.3f us/launch' % (d['value'] / 1e6, d['ms_per_step'] * 1e3, d['ms_per_step_min'] * 1e3, d['roofline']['kernel_us']))"
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('K=2000: %.1f M env-steps/s  wall %.3f us/step (min %.3f)  events %.3f us/launch' % (d['value'] / 1e6, d['ms_per_step'] * 1e3, d['ms_per_step_min'] * 1e3, d['roofline']['kernel_us']))"
