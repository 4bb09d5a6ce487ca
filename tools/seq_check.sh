#!/bin/bash
# GPU box: bench runs back to back, each must print its line (a hand-off that times out makes bench.py fail loudly).
# usage: bash tools/seq_check.sh [rounds] [bench args...]
R=${1:-5}; shift
fails=0
for i in $(seq $R); do
  for k in "20 5" "2000 200"; do
    set -- $k "${@:3}"
    CZ_CHAIN_DEBUG=1 timeout 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline "${@:3}" > /tmp/seq.json 2> /tmp/seq.err
    if grep -q "gave up" /tmp/seq.err; then fails=$((fails+1)); echo "K=$1 round $i FAILED: $(grep 'gave up' /tmp/seq.err | tail -1 | cut -c150-260)";
    else python - <<PY
import json
d=json.loads(open("/tmp/seq.json").readline())
print("K=$1 round $i ok: %.1f M  wall %.3f us  interval %.3f us" % (d["value"]/1e6, d["ms_per_step"]*1e3, d["roofline"]["kernel_us"]))
PY
    fi
  done
done
echo "failures: $fails"
