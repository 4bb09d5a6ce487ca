#!/bin/bash
# GPU box: parity suite + the two bench lines (driver's K = 20 and the default K = 2000).  usage: bash tools/gpu_check.sh TAG [pytest args]
TAG=${1:-x}; shift
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -12 > gpurun_out/r02/gpu_tests_$TAG.log
one() { python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$1: %.1f M env-steps/s  wall %.3f us/step (min %.3f max %.3f)  events %.3f us/launch  frac %.4f  fused %.1f M' % (d['value'] / 1e6, d['ms_per_step'] * 1e3, d['ms_per_step_min'] * 1e3, d['ms_per_step_max'] * 1e3, d['roofline']['kernel_us'], d['roofline']['frac'], d.get('fused_rollout', {}).get('env_steps_per_s_per_gpu', 0) / 1e6))"; }
python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | head -1 | tee gpurun_out/r02/bench_${TAG}_k2000.json | one K2000 > gpurun_out/r02/bench_$TAG.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | head -1 | tee gpurun_out/r02/bench_${TAG}_k20.json | one K20 >> gpurun_out/r02/bench_$TAG.txt
cat gpurun_out/r02/gpu_tests_$TAG.log gpurun_out/r02/bench_$TAG.txt
