#!/bin/bash
# same-box A/B of two builds of the library: tools/ab.sh <libA> <libB> [bench_configs args]
# (CZ_LIB selects the shared object the ctypes binding loads)
A=$1; B=$2; shift 2
for rep in 1 2; do
  for lib in "$A" "$B"; do
    echo "== $lib (rep $rep)"
    CZ_LIB=$lib python tools/bench_configs.py "$@" 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    try: d = json.loads(line)
    except Exception: continue
    print('  %-28s per-step %7.1f M/s (%7.2f us)   fused %7.1f M/s' % (d['case'][:28], d['per_step_env_steps_per_s']/1e6, d['per_step_us'], d['fused_env_steps_per_s']/1e6))
"
  done
done
