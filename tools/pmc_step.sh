#!/bin/bash
# usage: bash tools/pmc_step.sh TAG   -- PMC instruction counts + timing of the bench step kernel (GPU box)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-x}
mkdir -p gpurun_out/pmc_$TAG gpurun_out/pmcn_$TAG
timeout 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_$TAG -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
echo "== with obs"; python3 tools/pmc_summary.py gpurun_out/pmc_$TAG '3, 0>'
timeout 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmcn_$TAG -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-obs > /dev/null 2>&1
echo "== no obs"; python3 tools/pmc_summary.py gpurun_out/pmcn_$TAG '3, 0>'
python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value']/1e6, 'M/s  ms_per_step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'])"
