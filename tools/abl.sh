#!/bin/bash
# GPU box, diagnostic build: dynamic instruction count of each phase = difference between successive CZ_STOP truncations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CZ_LIB=$GRAFT_REPO_ROOT/cooking_zoo_amd/csrc/libcookingzoo_hip_ablate.so
for m in 1 2 3 4 5 6 7 9; do
  mkdir -p gpurun_out/stop$m
  CZ_CHAIN=0 CZ_STOP=$m timeout 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/stop$m -- python3 tools/step_loop.py 30 > /dev/null 2>&1
  echo "stop=$m $(python3 tools/pmc_summary.py gpurun_out/stop$m '3, 0>' brief)"
done
