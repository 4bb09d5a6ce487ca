for rep in 1 2; do
for cfg in "CZ_GRAPHS=1 CZ_RING_PREFIX=0" "CZ_GRAPHS=1 CZ_RING_PREFIX=2" "CZ_GRAPHS=1 CZ_RING_PREFIX=4" "CZ_GRAPHS=0 CZ_RING_PREFIX=0"; do
  echo -n "$cfg: "; env $cfg python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.1f M  ms/step %.3f us  min %.1f max %.1f' % (d['value']/1e6, d['ms_per_step']*1e3, d['value_min']/1e6, d['value_max']/1e6))"
done; done
