#!/bin/bash
# experiment: the headline leg by batch size for several libraries (LIBS=...), interleaved
for n in ${SIZES:-4096 6144 8192 10240 12288 16384}; do for lib in $LIBS; do
  CZ_LIB=$lib timeout 300 python3 bench.py --envs $n --steps 400 --warmup 40 --repeats 10 --no-extras --no-cpu-baseline 2> /dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%7d envs  %-40s %8.2f us per launch  %8.1f M env-steps/s' % ($n, '$lib'.split('/')[-2], d['roofline']['kernel_us'], d['value'] / 1e6))
"
done; done
