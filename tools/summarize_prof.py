#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV to the rows of this repo's kernels (k_*, templated ones included), as a
small CSV.  usage: summarize_prof.py <kernel_stats.csv> <out.csv>"""
import collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kname import kname

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    n = kname(r["Name"])
    if n is None:
        continue
    a = acc.setdefault(n, {"calls": 0, "total": 0.0, "min": 1e30, "max": 0.0})   # template instantiations of one kernel pool
    a["calls"] += int(r["Calls"]); a["total"] += float(r["TotalDurationNs"])
    a["min"] = min(a["min"], float(r["MinNs"])); a["max"] = max(a["max"], float(r["MaxNs"]))
tot = sum(a["total"] for a in acc.values()) or 1.0
with open(sys.argv[2], "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,total_ms,share_of_svo_kernels\n")
    for n, a in sorted(acc.items(), key=lambda kv: -kv[1]["total"]):
        f.write("%s,%d,%.2f,%.2f,%.2f,%.3f,%.4f\n" % (n, a["calls"], a["total"] / a["calls"] / 1e3, a["min"] / 1e3, a["max"] / 1e3, a["total"] / 1e6, a["total"] / tot))
print(open(sys.argv[2]).read())
