#!/bin/bash
# GPU box: the cut libraries of tools/phase_cut.py under rocprofv3 - instruction counters (two passes) and a kernel trace per cut.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r05/phase_cut; rm -rf $O; mkdir -p $O
C=cooking_zoo_amd/csrc
[ -f $C/cuts/libcz_cut_1.so ] || python3 tools/phase_cut.py make > $O/make.log 2>&1
cp $C/cuts/cuts.json $O/
CZ_LIB=$C/libcookingzoo_hip_mark.so timeout 120 python3 tools/phase_cut_run.py warm $O/warm.npy
for i in 0 1 2 3 4 5 6 7 full; do
  lib=$C/cuts/libcz_cut_$i.so; [ $i = full ] && lib=$C/libcookingzoo_hip_mark.so
  mkdir -p $O/cut_$i/a $O/cut_$i/b $O/cut_$i/t
  CZ_LIB=$lib timeout 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/cut_$i/a -- python3 tools/phase_cut_run.py run $O/warm.npy 64 > /dev/null 2>&1
  CZ_LIB=$lib timeout 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/cut_$i/b -- python3 tools/phase_cut_run.py run $O/warm.npy 64 > /dev/null 2>&1
  CZ_LIB=$lib timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/cut_$i/t -- python3 tools/phase_cut_run.py run $O/warm.npy 2000 > /dev/null 2>&1
  echo "cut $i rc=$?"
done
python3 tools/phase_cut.py table $O | tee $O/phase_table.md
# the shipped library under the same driver, for the row "whole kernel, shipped build"
mkdir -p $O/ship/t
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ship/t -- python3 tools/phase_cut_run.py run $O/warm.npy 2000 > /dev/null 2>&1
find $O/ship -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-200
rm -f $O/warm.npy; find $O -name "*.csv" -size +2M -delete
