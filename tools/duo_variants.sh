#!/bin/bash
# GPU box: experiment variants of the two-waves-per-env kernel (CZ_DUO=1 CZ_STOP=v, see cz_duo.h) against the ordinary kernel
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
{
for rep in 1 2; do
  echo "== ordinary kernel, rep $rep"; CZ_DUO=0 timeout 200 python3 tools/mode_timing.py
  for v in ${VARIANTS:-1 2 3 0}; do
    echo "== CZ_DUO=1 variant $v, rep $rep"; CZ_DUO=1 CZ_STOP=$v timeout 200 python3 tools/mode_timing.py
  done
done
for v in ${VARIANTS:-1 2 3 0}; do
  echo "== parity, variant $v"; CZ_DUO=1 CZ_STOP=$v timeout 600 python -m pytest tests -m gpu -x -q -k "parity or rollout_actions or spawn_device" 2>&1 | tail -3
done
} > $O/duo_variants_$TAG.txt 2>&1
cat $O/duo_variants_$TAG.txt
