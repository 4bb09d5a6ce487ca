#!/bin/bash
# GPU box: a parity subset + same-box A/B of .ab/prev (the round-3 tree, built; git-ignored) against this tree.
# usage: bash tools/r04_check.sh TAG ["pytest -k expression"]
TAG=${1:-x}; K=${2:-"rollout or parity or api"}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -8 > $O/tests_$TAG.log
for rep in 1 2; do
  timeout 200 python3 tools/mode_timing.py > $O/modes_${TAG}_$rep.txt 2>&1
  [ -d .ab/prev ] && (cd .ab/prev && timeout 200 python3 ../../tools/mode_timing.py) > $O/modes_prev_${TAG}_$rep.txt 2>&1
done
if [ -d .ab/prev ]; then CZ_CHAIN=0 timeout 900 bash tools/ab_trees.sh .ab/prev . 2 --no-extras > $O/ab_$TAG.txt 2>&1; fi
tail -n 20 $O/tests_$TAG.log $O/modes_${TAG}_*.txt $O/modes_prev_${TAG}_*.txt $O/ab_$TAG.txt
