#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per svo kernel from a counter_collection.csv."""
import csv, sys, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kname import kname
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = kname(r["Kernel_Name"])
    if n is None: continue
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
print("kernel," + ",".join(names))
for k, d in acc.items():
    print(k + "," + ",".join("%.4g" % (sum(d[c]) / len(d[c])) if c in d else "" for c in names))
