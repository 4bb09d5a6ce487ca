#!/bin/bash
# GPU box: everything profiles/r04/ holds, from one box.  usage: bash tools/collect_r04.sh   (results under gpurun_out/r04_final/)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04_final
rm -rf $O; mkdir -p $O
# 1. device-clock timelines (timeline library): boundary-ordered and overlapped, bench workload
python3 tools/timeline.py 4096 4000 0 > $O/timeline_ordered.json 2> $O/timeline.err
python3 tools/timeline.py 4096 4000 1 > $O/timeline_overlapped.json 2>> $O/timeline.err
# 2. rocprofv3: kernel trace + stats, traffic and instruction counters of the headline workload (tools/collect_profiles.sh)
bash tools/collect_profiles.sh r04final > $O/collect_profiles.log 2>&1
cp -r gpurun_out/prof_r04final $O/prof
# 3. large batches: counters + traces of configs 3, 5 and the config-4 shard
bash tools/pmc_large.sh r04final cfg3 cfg5 cfg4_shard > $O/pmc_large.log 2>&1
cp gpurun_out/pmc_large_r04final/summary.json $O/large_batch_counters.json
# 4. instruction counts / launch time per action stream, the cooking workload's phases, this tree against the round-3 tree
bash tools/interact_probe.sh > $O/interact_probe.txt 2>&1
for rep in 1 2; do echo "== this tree"; python3 tools/mode_timing.py; [ -d .ab/prev ] && (echo "== round-3 tree (.ab/prev)"; cd .ab/prev && python3 ../../tools/mode_timing.py); done > $O/mode_timing.txt 2>&1
CASES="cfg5 cfg3 cfg4_shard" bash tools/ab_configs.sh > /dev/null 2>&1; cp gpurun_out/r04/ab_configs.txt $O/ab_configs_vs_r03.txt
# 5. the compact observation, the closed loops, host-array steps, issue modes of a K-step region
python3 tools/compact_sizes.py 4096 16384 32768 65536 131072 > $O/compact_sizes.txt 2>&1
python3 tools/closed_loop_parts.py > $O/closed_loop_parts.txt 2>&1
python3 tools/host_step_latency.py > $O/host_step_latency.txt 2>&1
bash tools/bench_modes.sh > /dev/null 2>&1; cp gpurun_out/r04/bench_modes.txt $O/bench_modes.txt
# 6. the bench lines: default, the driver's K = 20, boundary-ordered
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
CZ_CHAIN=0 python3 bench.py --no-extras --no-cpu-baseline > $O/bench_ordered.json 2> /dev/null
ls -la $O
