#!/bin/bash
# GPU box: everything profiles/r03/ holds, from one box.  usage: bash tools/collect_r03.sh   (results under gpurun_out/r03_final/)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r03_final
rm -rf $O; mkdir -p $O
# 1. device-clock timelines (timeline library): boundary-ordered and overlapped, bench workload; stay / no-obs variants
python3 tools/timeline.py 4096 4000 0 > $O/timeline_ordered.json 2> $O/timeline.err
python3 tools/timeline.py 4096 4000 1 > $O/timeline_overlapped.json 2>> $O/timeline.err
for cfg in "stay 400 1" "stay 400 0" "random 400 0" "random 1073741824 1"; do set -- $cfg; python3 tools/timeline.py 4096 1000 0 $1 $2 $3 2>/dev/null; done > $O/timeline_variants.jsonl
# 2. rocprofv3: kernel trace + stats, traffic and instruction counters of the headline workload (tools/collect_profiles.sh)
bash tools/collect_profiles.sh r03final > $O/collect_profiles.log 2>&1
cp -r gpurun_out/prof_r03final $O/prof
# 3. large batches: counters + traces of configs 3, 5 and the config-4 shard; size sweep; fused-length sweep; write-through A/B
bash tools/pmc_large.sh r03final cfg3 cfg5 cfg4_shard > $O/pmc_large.log 2>&1
cp gpurun_out/pmc_large_r03final/summary.json $O/large_batch_counters.json
python3 tools/size_sweep.py 1 > $O/size_sweep.txt 2>&1
python3 tools/size_sweep.py 0 32768 65536 131072 >> $O/size_sweep.txt 2>&1
python3 tools/fuse_sweep.py 131072 > $O/fuse_sweep.txt 2>&1
python3 tools/fuse_sweep.py 65536 >> $O/fuse_sweep.txt 2>&1
for wt in 0 1 2; do echo "CZ_WT=$wt"; CZ_WT=$wt timeout 400 python3 tools/bench_configs.py cfg2_16k cfg4_shard cfg3 cfg5 2>/dev/null | cut -c1-420; done > $O/wt_ab.txt
(echo "== plain stores"; CZ_WT=0 python3 tools/fuse_sweep.py 4096 4,8,16,32,64,128 1024; echo "== streaming stores"; CZ_WT=2 python3 tools/fuse_sweep.py 4096 4,8,16,32,64,128 1024; echo "== no observation"; python3 tools/fuse_sweep.py 4096 8,32,128 1024 noobs) > $O/fuse_sweep_4096.txt 2>&1
# 4. instruction counts per action stream, launch time per action stream, phase stamps
bash tools/interact_probe.sh > $O/interact_probe.txt 2>&1
python3 tools/mode_timing.py > $O/mode_timing.txt 2>&1
for m in stay random; do python3 tools/phase_profile.py 4096 $m; done > $O/phase_profile.txt 2>&1
python3 tools/cook_profile.py 200 > $O/cook_profile.txt 2>&1
python3 tools/rot_probe.py > $O/rot_probe.txt 2>&1
# 5. the bench lines: default, the driver's K = 20, boundary-ordered
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
CZ_CHAIN=0 python3 bench.py --no-extras --no-cpu-baseline > $O/bench_ordered.json 2> /dev/null
python3 tools/facade_latency.py > $O/facade_latency.txt 2>&1
ls -la $O
