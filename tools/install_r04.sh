#!/bin/bash
# build container: copy what tools/collect_r04.sh produced (gpurun_out/r04_final, gpurun_out/pmc_large_r04final) into profiles/r04/
cd "$(dirname "$0")/.."
S=gpurun_out/r04_final; D=profiles/r04
mkdir -p $D
for f in timeline_ordered.json timeline_overlapped.json large_batch_counters.json interact_probe.txt mode_timing.txt ab_configs_vs_r03.txt \
         compact_sizes.txt closed_loop_parts.txt host_step_latency.txt bench_modes.txt bench_default.json bench_k20.json bench_ordered.json; do
  [ -s $S/$f ] && cp $S/$f $D/$f || echo "missing $f"
done
P=$S/prof
cp $P/pmc_summary.json $D/pmc_summary.json
cp $P/trace_gaps.txt $D/trace_gaps.txt; cp $P/trace_gaps_ordered.txt $D/trace_gaps_ordered.txt
cp $P/bench_under_trace.json $D/; cp $P/bench_under_trace_ordered.json $D/
cp $(ls -t $P/trace/runc/*kernel_stats.csv | head -1) $D/kernel_stats.csv
cp $(ls -t $P/trace_ordered/runc/*kernel_stats.csv | head -1) $D/kernel_stats_ordered.csv
for c in cfg3 cfg5 cfg4_shard; do cp $(ls -t gpurun_out/pmc_large_r04final/${c}_trace/runc/*kernel_stats.csv | head -1) $D/kernel_stats_$c.csv; done
python3 - <<'PY'
import json
s = json.load(open("gpurun_out/r04_final/prof/pmc_summary.json"))["step"]
alg = 4655 * 4096
t = (2 * s["FETCH_SIZE_KB_per_launch"] + s["WRITE_SIZE_KB_per_launch"]) * 1024
json.dump({"k_step_4096": t,
           "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh, CZ_CHAIN=0), bytes per launch = "
                   "(2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE counts 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM), "
                   "WRITE_SIZE is exact for 16-B-per-lane stores",
           "FETCH_SIZE_KB": s["FETCH_SIZE_KB_per_launch"], "WRITE_SIZE_KB": s["WRITE_SIZE_KB_per_launch"], "algorithmic_bytes": alg,
           "ratio": t / alg, "dispatches_averaged": s["dispatches"],
           "_kernel": "cz::k_step<1,1,2,3,0>, launches ordered by launch boundaries"}, open("profiles/r04/traffic.json", "w"), indent=1)
print("traffic ratio", t / alg)
PY
python3 tools/kernel_metadata.py > $D/kernel_metadata.txt
ls $D | wc -l
