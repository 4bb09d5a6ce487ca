#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/instance_loop.py -> instructions per env-step of every kernel instance (issue_per_env_step.json:
what bench.py's `roofline.issue` blocks are computed from).

    python3 tools/pmc_instances.py DIR      # DIR/inst_<form>/ hold the counter_collection.csv files, DIR/inst_<form>.out the driver's line"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    root = sys.argv[1]
    out = {"_how": "rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES (one pass) and SQ_INSTS_SMEM SQ_INSTS_BRANCH "
                   "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR (another) over tools/instance_loop.py FORM, 4096 envs of the bench workload; per "
                   "wave (= per env) and launch, divided by the env steps a launch makes (tools/collect_round.sh)"}
    for line_file in sorted(glob.glob(os.path.join(root, "inst_*.out"))):
        m = re.search(r"instance=(\S+) steps_per_launch=(\d+)", open(line_file).read())
        if not m:
            continue
        inst, steps = m.group(1), int(m.group(2))
        d = line_file[:-4]
        want = inst.split("/")[0].replace(",", ", ")                 # "k_step<1,1,2,3,0>/cooking" -> the kernel name's "k_step<1, 1, 2, 3, 0>"
        agg = collections.defaultdict(list)
        for f in glob.glob(d + "*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if want in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]) / (int(r["Grid_Size"]) // 64))
        if "SQ_INSTS_SALU" not in agg:
            continue
        per = {k: sum(v) / len(v) / steps for k, v in agg.items()}
        out[inst] = {"salu": round(per["SQ_INSTS_SALU"], 1), "valu": round(per["SQ_INSTS_VALU"], 1), "lds": round(per.get("SQ_INSTS_LDS", 0.0), 1),
                     "smem": round(per.get("SQ_INSTS_SMEM", 0.0), 1), "branch": round(per.get("SQ_INSTS_BRANCH", 0.0), 1),
                     "vmem_rd": round(per.get("SQ_INSTS_VMEM_RD", 0.0), 1), "vmem_wr": round(per.get("SQ_INSTS_VMEM_WR", 0.0), 1),
                     "steps_per_launch": steps, "dispatches": len(agg["SQ_INSTS_SALU"])}
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
