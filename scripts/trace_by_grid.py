#!/usr/bin/env python
"""rocprofv3 --kernel-trace writes one row per dispatch; --stats averages a kernel NAME over all its dispatches.  bench.py runs
the batch as one launch sequence AND as two half-batch sequences, so one name covers launches of two sizes: this script keeps
the per-(kernel, grid size) averages of a trace directory in <dir>/trace_by_grid.csv (the per-dispatch CSV itself is too big
to keep).  usage: trace_by_grid.py <trace dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

directory = sys.argv[1]
acc = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            dur = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            key = (row["Kernel_Name"], int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1) * int(row.get("Grid_Size_Z", 1) or 1))
            a = acc[key]
            a[0] += 1
            a[1] += dur
            a[2] = min(a[2], dur)
            a[3] = max(a[3], dur)
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])[:60]
with open(os.path.join(directory, "trace_by_grid.csv"), "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "GridSize", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for (name, grid), (n, total, lo, hi) in rows:
        w.writerow([name, grid, n, int(total), round(total / n, 1), int(lo), int(hi)])
print(f"{len(acc)} (kernel, grid) groups -> {os.path.join(directory, 'trace_by_grid.csv')}")
