#!/bin/bash
# usage: bash tools/pmc_lds.sh TAG   -- LDS pipe occupancy and bank conflicts of the bench step kernel (GPU box)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-x}
mkdir -p gpurun_out/pmcl_$TAG
timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcl_$TAG -- python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcl_$TAG '3, 0>'
