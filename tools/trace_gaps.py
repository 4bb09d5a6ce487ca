#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: kernel duration vs start-to-start interval of consecutive k_step launches."""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
a = np.array(rows[50:])
dur = a[:, 1] - a[:, 0]
gap = a[1:, 0] - a[:-1, 1]
itv = a[1:, 0] - a[:-1, 0]
print(f"launches {len(a)}  duration median {np.median(dur)/1e3:.2f} us  mean {dur.mean()/1e3:.2f} us | gap median {np.median(gap)/1e3:.2f} us | interval median {np.median(itv)/1e3:.2f} us")
