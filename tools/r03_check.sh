#!/bin/bash
# GPU box: quick parity subset + instruction counts per mode + same-box A/B bench of .ab/prev against this tree.
# (.ab/prev = a built copy of the tree to compare with, git-ignored: `git worktree add .ab/prev <commit> && make -C .ab/prev/cooking_zoo_amd/csrc`;
#  without it only this tree is measured.  For A/Bs of single kernels see tools/ab_libs.sh / tools/ab_overlap.sh with CZ_LIB.)
# usage: bash tools/r03_check.sh TAG [pytest -k expression]
TAG=${1:-x}; K=${2:-"parity or rollout or fuzz or api"}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -8 > gpurun_out/r03/tests_$TAG.log
bash tools/interact_probe.sh > gpurun_out/r03/ip_$TAG.txt 2>&1
python3 tools/mode_timing.py > gpurun_out/r03/modes_$TAG.txt 2>&1; [ -d .ab/prev ] && (cd .ab/prev && python3 ../../tools/mode_timing.py) > gpurun_out/r03/modes_prev_$TAG.txt 2>&1
if [ -d .ab/prev ]; then CZ_CHAIN=0 bash tools/ab_trees.sh .ab/prev . 2 > gpurun_out/r03/ab_$TAG.txt 2>&1; fi
cat gpurun_out/r03/modes_$TAG.txt gpurun_out/r03/modes_prev_$TAG.txt gpurun_out/r03/tests_$TAG.log gpurun_out/r03/ip_$TAG.txt gpurun_out/r03/ab_$TAG.txt
