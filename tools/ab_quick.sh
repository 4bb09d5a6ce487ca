#!/bin/bash
# same-box A/B of libraries (paths in $LIBS, default the shipped one): mode_timing (stay / random / rich launch times) and the bench's
# ordered pass + legs, two rounds, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in ${LIBS:-cooking_zoo_amd/csrc/libcookingzoo_hip.so}; do echo "== $lib"; CZ_LIB=$lib timeout 200 python3 tools/mode_timing.py; CZ_LIB=$lib timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.1f M' % (d['value']/1e6), 'kernel_us %.3f' % d['roofline']['kernel_us'], 'cook %.3f' % d['cooking_policy']['us_per_launch'], 'closed %.3f' % d['closed_loop']['us_per_step'], 'fused %.3f' % (d['fused_rollout']['ms_per_step']*1e3), 'fused_compact %.3f' % (d['fused_compact']['ms_per_step']*1e3), 'ring_fused %.3f' % d['ring_fused']['K=2000']['us_per_step'])
"; done; done
