for n in ${SIZES:-49152 65536 98304}; do for wt in -1 0 1 2; do
  if [ $wt = -1 ]; then unset CZ_WT; else export CZ_WT=$wt; fi
  timeout 300 python3 bench.py --envs $n --steps 400 --warmup 40 --repeats 10 --no-extras --no-cpu-baseline 2> /dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print('$n envs, CZ_WT=$wt: %8.2f us per launch  %8.1f M env-steps/s  hbm frac %.3f' % (r['kernel_us'], d['value'] / 1e6, r['frac']))
"
done; done
