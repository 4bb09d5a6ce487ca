#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch and per wave for one kernel (substring match)."""
import collections
import csv
import glob
import sys


def main():
    d, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_step"
    brief = len(sys.argv) > 3
    agg = collections.defaultdict(list)
    waves = None
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                waves = int(r["Grid_Size"]) // 64
    if brief:
        print(" ".join(f"{k}={sum(v) / len(v) / max(waves or 1, 1):.0f}" for k, v in sorted(agg.items())))
        return
    print("kernel;counter;dispatches;mean_per_dispatch;per_wave")
    for k, v in sorted(agg.items()):
        m = sum(v) / len(v)
        print(f"{pat};{k};{len(v)};{m:.1f};{m / max(waves or 1, 1):.1f}")


if __name__ == "__main__":
    main()
