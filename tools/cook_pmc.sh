#!/bin/bash
# GPU box: instructions per wave and step of the one-step kernel under the heuristic agent's action stream and under uniform
# random actions on the same worlds (separate --pmc pass each)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04/cook_pmc; rm -rf $O; mkdir -p $O/cook $O/random
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/cook -- python3 tools/cook_ring.py > $O/cook.out 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/random -- python3 tools/cook_ring.py random > $O/random.out 2>&1
for w in cook random; do echo "== $w"; python3 tools/pmc_summary.py $O/$w '3, 0>'; done | tee gpurun_out/r04/cook_pmc.txt
