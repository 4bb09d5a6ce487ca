#!/bin/bash
# GPU box, ablation build (make ablate): dynamic instruction count of each phase = difference between successive CZ_STOP
# truncations.   usage: bash tools/abl.sh [steps=400] [random|zero]
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
export CZ_LIB=$GRAFT_REPO_ROOT/cooking_zoo_amd/csrc/libcookingzoo_hip_ablate.so
STEPS=${1:-400}; MODE=${2:-random}
for m in 1 2 3 4 5 6 7; do
  mkdir -p gpurun_out/stop$m; rm -rf gpurun_out/stop$m/*
  CZ_STOP=$m timeout 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/stop$m -- python3 tools/step_loop.py $STEPS 4096 $MODE > /dev/null 2>&1
  echo "$MODE stop=$m $(python3 tools/pmc_summary.py gpurun_out/stop$m '3, 0>' brief)"
done
