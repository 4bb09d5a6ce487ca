#!/bin/bash
# same-box A/B of experiment libraries (CZ_LIB): launch time under three action streams + the bench legs
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in ${LIBS:-libcookingzoo_hip.so}; do echo "== $lib"; CZ_LIB=cooking_zoo_amd/csrc/$lib timeout 200 python3 tools/mode_timing.py; CZ_LIB=cooking_zoo_amd/csrc/$lib timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ordered', d['roofline']['kernel_us'], 'cook', d['cooking_policy']['us_per_launch'], 'closed', d['closed_loop']['us_per_step'], 'fused', d.get('fused_rollout',{}).get('ms_per_step'))
"; done; done
