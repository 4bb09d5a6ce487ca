#!/bin/bash
# GPU box: what profiles/rNN/ holds of rocprofv3, from one box.  usage: bash tools/collect_round.sh r06   (results under gpurun_out/r06/;
# issue_per_env_step.json and traffic.json are also put into profiles/r06/ on the box so that the last bench line reads them)
# Every profiled command is the program itself behind `--` and runs under `timeout`.
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
R=${1:?round directory name, e.g. r06}
O=gpurun_out/$R
rm -rf $O; mkdir -p $O/trace $O/trace_k20 $O/fetch $O/write
# 1. the bench line, then the same command under a kernel trace: the headline's launches ARE the roofline's kernel
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 400 --warmup 40 --repeats 5 --no-extras --no-cpu-baseline > $O/bench_under_trace.json 2> $O/trace.log
echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k20 -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_under_trace_k20.json 2> $O/trace_k20.log
echo "trace_k20 rc=$?"
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/trace_k20 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_k20.csv \;
# 2. HBM traffic of that kernel: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM / rocprofv3)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/step_loop.py 400 > /dev/null 2>&1
echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/step_loop.py 400 > /dev/null 2>&1
echo "write rc=$?"
# 3. instructions per env-step of every kernel instance the bench line quotes (two counter sets, one pass each)
for form in step step_compact rollout rollout_actions rollout_compact ring_in_place cooking; do
  mkdir -p $O/inst_${form}_a $O/inst_${form}_b
  timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/inst_${form}_a -- python3 tools/instance_loop.py $form > $O/inst_$form.out 2> $O/inst_$form.err
  timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/inst_${form}_b -- python3 tools/instance_loop.py $form > /dev/null 2>&1
  echo "inst $form rc=$?"
done
python3 tools/pmc_instances.py $O > $O/issue_per_env_step.json
python3 - <<PY
import csv, glob, json, collections
O="$O"
def pmc(d, pat):
    v=[]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]: v.append(float(r["Counter_Value"]))
    return (sum(v)/len(v), len(v)) if v else (None, 0)
f, w = pmc(O+"/fetch", "k_step<1, 1, 2, 3, 0>"), pmc(O+"/write", "k_step<1, 1, 2, 3, 0>")
if f[0] and w[0]:
    alg = 4655 * 4096
    t = (2 * f[0] + w[0]) * 1024
    json.dump({"k_step_4096": t, "FETCH_SIZE_KB": f[0], "WRITE_SIZE_KB": w[0], "algorithmic_bytes": alg, "ratio": t / alg,
               "dispatches_averaged": {"fetch": f[1], "write": w[1]}, "_kernel": "cz::k_step<1,1,2,3,0>, launches ordered by launch boundaries",
               "_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/collect_round.sh), bytes per launch = (2 x FETCH_SIZE + "
                       "WRITE_SIZE) x 1024: FETCH_SIZE counts 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact for "
                       "16-B-per-lane stores"}, open(O+"/traffic.json", "w"), indent=1)
PY
# 4. the bench line again now that this box's instruction counts exist (issue blocks filled in from this box)
mkdir -p profiles/$R && cp $O/issue_per_env_step.json profiles/$R/issue_per_env_step.json && cp $O/traffic.json profiles/$R/traffic.json 2>/dev/null
timeout 900 python3 bench.py > $O/bench_default_with_issue.json 2> /dev/null
rm -rf $O/trace $O/trace_k20 $O/fetch $O/write $O/inst_*_a $O/inst_*_b
ls -la $O; cat $O/kernel_stats.csv | cut -c1-200 | head -8; cat $O/issue_per_env_step.json
