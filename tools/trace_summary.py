#!/usr/bin/env python3
"""Per-launch durations of the svo kernels from a rocprofv3 kernel_trace.csv, grouped by (kernel, grid size)."""
import csv, sys, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kname import kname
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    n = kname(r["Kernel_Name"])
    if n is None: continue
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc.setdefault(key, []).append(d)
for k, v in acc.items():
    v = sorted(v)
    print("%-18s grid=%s,%s,%s  n=%d  median_us=%.1f  min=%.1f" % (k[0], k[1], k[2], k[3], len(v), v[len(v) // 2], v[0]))
