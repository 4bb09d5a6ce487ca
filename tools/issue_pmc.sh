#!/bin/bash
# GPU box: what the compute units are busy with during a 4096-env launch of the one-step kernel (bench workload, launch-boundary
# ordering): issue-cycle counters of the scalar unit, the vector ALUs, LDS and memory instructions, in separate --pmc passes.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04/issue_pmc; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_IFETCH"; do
  i=$((i + 1)); mkdir -p $O/p$i
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 tools/step_loop.py 400 > $O/p$i.out 2>&1
done
python3 tools/pmc_summary.py $O '3, 0>' | tee gpurun_out/r04/issue_pmc.txt
