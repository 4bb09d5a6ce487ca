#!/bin/bash
# GPU box: HBM traffic counters (separate --pmc passes, as the microarch guide prescribes) and kernel trace of the one-step and the
# fused kernel at large batches.  usage: bash tools/pmc_large.sh TAG case...   (cases of tools/bench_configs.py: cfg3 cfg5 cfg4_shard cfg2_64k)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
TAG=${1:-x}; shift
CASES=${@:-cfg3}
O=gpurun_out/pmc_large_$TAG
mkdir -p $O
for c in $CASES; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    d=$O/${c}_$(echo $pass | cut -d' ' -f1)
    mkdir -p $d
    timeout 300 rocprofv3 --pmc $pass --output-format csv -d $d -- python3 tools/bench_configs.py $c > $d/out.txt 2>&1
    echo "$c [$pass] rc=$?"
  done
  d=$O/${c}_trace; mkdir -p $d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_configs.py $c > $d/out.txt 2>&1
  echo "$c [trace] rc=$?"
done
python3 - <<PY
import csv, glob, collections, json, os
O = "$O"
res = {}
for d in sorted(glob.glob(O + "/*")):
    if not os.path.isdir(d): continue
    name = os.path.basename(d)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_step" in k:
                agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_step" in r["Name"]:
                res.setdefault(name, {})[r["Name"].split("(")[0][:60]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    for k, cs in agg.items():
        res.setdefault(name, {})[k] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()}
json.dump(res, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
