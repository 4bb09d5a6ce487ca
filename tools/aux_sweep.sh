#!/bin/bash
# cache-policy bits of the observation stores at large batches (experiment libraries libcz_aux<bits>.so: -DCZ_PLAIN_AUX=<bits>)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r03
for a in 0 2 3 18 19; do
  lib=cooking_zoo_amd/csrc/libcz_aux$a.so; [ $a = 0 ] && lib=cooking_zoo_amd/csrc/libcookingzoo_hip.so
  echo "== aux $a (CZ_WT=0)"; CZ_LIB=$lib CZ_WT=0 timeout 300 python3 tools/size_sweep.py 1 4096 32768 49152 65536 131072
done
