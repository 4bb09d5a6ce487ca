#!/bin/bash
# GPU box: build the library with extra compile flags and print the bench lines.  usage: bash tools/ab_flags.sh "<flags>" [env assignments...]
FLAGS="$1"; shift
cd cooking_zoo_amd/csrc && make clean >/dev/null && make -j4 CXXFLAGS="-O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=${PRELOAD:-14} $FLAGS" >/dev/null 2>&1; cd ../..
for c in 0 1; do
  env CZ_CHAIN=$c "$@" timeout 200 python bench.py --steps 2000 --warmup 200 --repeats 10 --no-cpu-baseline 2>/dev/null | head -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('flags [$FLAGS] CZ_CHAIN=$c: %.1f M  wall %.3f us/step (min %.3f)  events %.3f us/launch' % (d['value'] / 1e6, d['ms_per_step'] * 1e3, d['ms_per_step_min'] * 1e3, d['roofline']['kernel_us']))"
done
