cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0 1 3 7 15 31; do
  mkdir -p gpurun_out/abl$m
  CZ_DEBUG_SKIP=$m rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/abl$m -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-obs > /dev/null 2>&1
  echo "skip=$m"; python3 tools/pmc_summary.py gpurun_out/abl$m | tail -3
done
