#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py's workload.  usage: bash tools/collect_profiles.sh TAG
# (kernel trace + stats in one run; FETCH_SIZE and WRITE_SIZE in separate --pmc passes, as the microarch guide prescribes)
# The counter passes run with CZ_CHAIN=0: a --pmc run executes one kernel at a time in an order of its own, and an
# overlapped launch that is run before its predecessor waits for it until the hand-off deadline.
# Every profiled command runs under `timeout`: a profiler-side abort must not be able to hold the box.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-cur}
O=gpurun_out/prof_$TAG
TRACE_ARGS=${TRACE_ARGS:---steps 400 --warmup 40 --repeats 5 --no-extras}
PMC_ARGS="--steps 40 --warmup 10 --repeats 3 --no-extras"
mkdir -p $O/trace $O/fetch $O/write $O/insts
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py $TRACE_ARGS --no-cpu-baseline > $O/bench_under_trace.json 2> $O/trace.log
echo "trace rc=$?"
CZ_CHAIN=0 timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py $PMC_ARGS --no-cpu-baseline > /dev/null 2>&1
echo "fetch rc=$?"
CZ_CHAIN=0 timeout 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py $PMC_ARGS --no-cpu-baseline > /dev/null 2>&1
echo "write rc=$?"
CZ_CHAIN=0 timeout 240 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/insts -- python3 bench.py $PMC_ARGS --no-cpu-baseline > /dev/null 2>&1
echo "insts rc=$?"
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err
# the same trace with the overlap switched off (launch boundaries only): per-kernel durations that mean something on their own
mkdir -p $O/trace_ordered
CZ_CHAIN=0 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ordered -- python3 bench.py $TRACE_ARGS --no-cpu-baseline > $O/bench_under_trace_ordered.json 2> $O/trace_ordered.log
echo "trace_ordered rc=$?"
python3 tools/trace_gaps.py $O/trace_ordered > $O/trace_gaps_ordered.txt
python3 tools/trace_gaps.py $O/trace > $O/trace_gaps.txt
python3 - <<PY
import csv, glob, json, collections
O="$O"
def pmc(d, pat):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k:(sum(v)/len(v), len(v)) for k,v in agg.items()}
out={}
for pat,key in (("k_step<1, 1, 2, 3, 0>", "step"), ("k_step<1, 1, 2, 3, 1>", "fused")):
    f=pmc(O+"/fetch", pat); w=pmc(O+"/write", pat); i=pmc(O+"/insts", pat)
    out[key]={"FETCH_SIZE_KB_per_launch": f.get("FETCH_SIZE",(None,0))[0], "WRITE_SIZE_KB_per_launch": w.get("WRITE_SIZE",(None,0))[0],
              "insts": {k:v[0] for k,v in i.items()}, "dispatches": {"fetch": f.get("FETCH_SIZE",(0,0))[1], "write": w.get("WRITE_SIZE",(0,0))[1]}}
json.dump(out, open(O+"/pmc_summary.json","w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat $O/trace_gaps.txt $O/trace_gaps_ordered.txt; find $O/trace $O/trace_ordered -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-220
