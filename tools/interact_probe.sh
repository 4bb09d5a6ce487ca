#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
mkdir -p gpurun_out/ip
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d gpurun_out/ip -- python3 tools/interact_probe.py 2>/dev/null | grep -v "RCCL\|version\|Hostname\|Librccl"
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/ip/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "3, 0>" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
n = len(ids) // 3
for k, mode in enumerate(("stay", "random", "bump")):
    sel = ids[k * n + 5:(k + 1) * n]
    avg = {c: sum(rows[i][c] for i in sel) / len(sel) / 4096 for c in rows[ids[0]]}
    print(mode, {c: round(v, 1) for c, v in avg.items()})
PY
