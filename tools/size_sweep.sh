#!/bin/bash
# GPU box: the headline leg (one launch of cz::k_step<1,1,2,3,0> per env step, graph replay) by batch size - where the launch reaches the HBM roof.
# usage: bash tools/size_sweep.sh r06   -> gpurun_out/r06/size_sweep.txt
R=${1:?round directory name}
O=gpurun_out/$R; mkdir -p $O
: > $O/size_sweep.txt
for n in 512 1024 2048 4096 6144 8192 12288 16384 24576 32768 65536 131072; do
  timeout 300 python3 bench.py --envs $n --steps 400 --warmup 40 --repeats 10 --no-extras --no-cpu-baseline 2> /dev/null | python3 -c "
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print(f\"{$n:7d} envs: {r['kernel_us']:8.2f} us per launch (HIP events)  {d['value'] / 1e6:8.1f} M env-steps/s  hbm frac {r['frac']:.3f}  issue frac {r['issue']['frac'] if r.get('issue') else float('nan'):.3f}  output-only launch {r['output_only_launch_us']:.2f} us\")
" >> $O/size_sweep.txt
done
cat $O/size_sweep.txt
