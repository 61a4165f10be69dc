#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV to the rows of this repo's kernels (k_*), as a small CSV.
usage: summarize_prof.py <kernel_stats.csv> <out.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ours = [r for r in rows if r["Name"].startswith("k_")]
tot = sum(float(r["TotalDurationNs"]) for r in ours) or 1.0
with open(sys.argv[2], "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,total_ms,share_of_svo_kernels\n")
    for r in sorted(ours, key=lambda r: -float(r["TotalDurationNs"])):
        f.write("%s,%s,%.2f,%.2f,%.2f,%.3f,%.4f\n" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                                                   float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["TotalDurationNs"]) / tot))
print(open(sys.argv[2]).read())
