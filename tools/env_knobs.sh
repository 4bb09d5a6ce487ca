#!/bin/bash
# GPU box: does a runtime knob change the steady-state launch time?  (bench K = 2000, 10 repeats each)
# Every run under `timeout`: ROC_SYSTEM_SCOPE_SIGNAL=0 never came back on this pool (dropped from the list).
# Measured in round 2: HIP_FORCE_DEV_KERNARG=0/1 no effect; AMD_OPT_FLUSH=0 8.1 us per launch (slower); the rest no effect.
one() { timeout 120 python bench.py --steps 2000 --warmup 200 --repeats 10 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-40s %.1f M env-steps/s  %.3f us/step  events %.3f us/launch' % ('$1', d['value'] / 1e6, d['ms_per_step'] * 1e3, d['roofline']['kernel_us']))"; }
one default
HIP_FORCE_DEV_KERNARG=1 one HIP_FORCE_DEV_KERNARG=1
HIP_FORCE_DEV_KERNARG=0 one HIP_FORCE_DEV_KERNARG=0
AMD_OPT_FLUSH=0 one AMD_OPT_FLUSH=0
ROC_USE_FGS_KERNARG=0 one ROC_USE_FGS_KERNARG=0
GPU_MAX_HW_QUEUES=1 one GPU_MAX_HW_QUEUES=1
one default
