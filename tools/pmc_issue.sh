#!/bin/bash
# usage: bash tools/pmc_issue.sh TAG   -- who uses the issue slots of the step kernel (GPU box); two SQ passes of 8 counters
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-x}
mkdir -p gpurun_out/pmci1_$TAG gpurun_out/pmci2_$TAG
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/r02/sq_counters.txt
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmci1_$TAG -- python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmci1_$TAG '3, 0>'
timeout 240 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmci2_$TAG -- python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmci2_$TAG '3, 0>'
