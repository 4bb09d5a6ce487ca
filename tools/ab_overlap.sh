#!/bin/bash
# same-box A/B of experiment libraries (CZ_LIB) on the overlapped launches: bench value at K = 2000 and K = 20, launch interval
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in ${LIBS:-libcookingzoo_hip.so}; do for k in "2000 200" "20 5"; do set -- $k
  CZ_LIB=cooking_zoo_amd/csrc/$lib timeout 300 python3 bench.py --steps $1 --warmup $2 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib K=$1 value %.1f M  overlapped interval %.3f us  ordered %.3f us' % (d['value']/1e6, r.get('overlapped_launch_interval_us') or 0, r['kernel_us']))
"; done; done; done
