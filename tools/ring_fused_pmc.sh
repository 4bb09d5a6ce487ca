#!/bin/bash
# GPU box: HBM traffic and duration of the fused ring launches (cz_set_ring_fused): do the rows really stay in the L2s?
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}
O=gpurun_out/r04/ring_fused_pmc; rm -rf $O; mkdir -p $O/w $O/f $O/t
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 tools/ring_fused_run.py 2000 5 > $O/w.out 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 tools/ring_fused_run.py 2000 5 > $O/f.out 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 tools/ring_fused_run.py 2000 5 > $O/t.out 2>&1
{
echo "fused ring runs of 2000 steps over a 64-slot ring (4096 envs): 32 launches of up to 64 steps per run; per launch (mean) and per env-step:"
python3 tools/pmc_summary.py $O '3, 2>'
python3 - "$O" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "3, 2>" in r["Name"]:
            print("kernel trace: %s calls, %.1f us per launch on average (62.5 steps per launch on average: %.3f us per step)" % (
                r["Calls"], float(r["AverageNs"]) / 1e3, float(r["AverageNs"]) / 1e3 / 62.5))
PY
echo "(WRITE_SIZE / FETCH_SIZE in KB per launch; a launch of 64 steps computes and stores 64 x 18.2 MB = 1165 MB of observation rows)"
} | tee gpurun_out/r04/ring_fused_pmc.txt
