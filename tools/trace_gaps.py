#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: per kernel variant of the step kernel, the duration of a launch against the
start-to-start interval of consecutive launches, and how many launches were in flight at once.

Ordinary launches (and graph replays) are ordered by launch boundaries: gaps >= 0, interval = duration + gap.  Overlapped
launches (removed in round 5) were resident two at a time: a kernel starts while its predecessor still runs and its waves wait,
env by env, for the predecessor's - so its duration is about twice the interval and says nothing about the rate; the
interval (= region time / launches, what bench.py measures with HIP events) does."""
import csv
import glob
import sys

import numpy as np

groups = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_step" not in name:
            continue
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
        chained = "k_step_chain" in name
        fused = not chained and len(args) > 4 and args[4] == "true"
        key = "fused rollout" if fused else ("one step, overlapped launches" if chained else "one step, boundary-ordered launches")
        groups.setdefault(key, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for key, rows in sorted(groups.items()):
    rows.sort()
    a = np.array(rows[min(50, len(rows) // 4):])
    if len(a) < 3:
        continue
    dur = a[:, 1] - a[:, 0]
    gap = a[1:, 0] - a[:-1, 1]
    itv = a[1:, 0] - a[:-1, 0]
    # launches in flight at the start of each launch (itself included)
    ends = np.sort(a[:, 1])
    inflight = np.arange(1, len(a) + 1) - np.searchsorted(ends, a[:, 0], side="right")
    print(f"{key}: launches {len(a)}  duration median {np.median(dur) / 1e3:.2f} us  mean {dur.mean() / 1e3:.2f} us | "
          f"gap median {np.median(gap) / 1e3:.2f} us | start-to-start interval median {np.median(itv) / 1e3:.2f} us | "
          f"in flight at a launch's start: median {int(np.median(inflight))}, max {int(inflight.max())}")
