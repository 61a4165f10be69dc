#!/usr/bin/env python3
"""Per-kernel launch statistics from a rocprofv3 kernel_trace.csv with the run's first steps left out, so that the
averages are the steady state the bench line reports (the first frames run the un-speculated FAST threshold and first-touch
everything).  Every launch of this repo's kernels (k_*, templated ones included) is counted.

usage: trace_stats.py <kernel_trace.csv> <out.csv> [--skip-steps S --total-steps T]
  --skip-steps S --total-steps T : of each kernel's launches, sorted by start time, the first S/T are dropped
                                   (a kernel launched n times per step is launched n*T times in the run)
The dropped launches are written as a second block of rows (kernel name + ":first_steps") so nothing is hidden."""
import argparse, collections, csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _kname import kname

ap = argparse.ArgumentParser()
ap.add_argument("trace"); ap.add_argument("out")
ap.add_argument("--skip-steps", type=int, default=0); ap.add_argument("--total-steps", type=int, default=0)
a = ap.parse_args()
launch = collections.OrderedDict()
for r in csv.DictReader(open(a.trace)):
    n = kname(r["Kernel_Name"])
    if n is None:
        continue
    launch.setdefault(n, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows, first = [], []
for n, v in launch.items():
    v.sort()
    cut = (len(v) * a.skip_steps) // a.total_steps if a.total_steps > 0 else 0
    for tag, part, dst in (("", v[cut:], rows), (":first_steps", v[:cut], first)):
        if part:
            d = [x[1] for x in part]
            dst.append((n + tag, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3, sum(d) / 1e6))
tot = sum(r[5] for r in rows) or 1.0
with open(a.out, "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,total_ms,share_of_svo_kernels\n")
    for r in sorted(rows, key=lambda r: -r[5]):
        f.write("%s,%d,%.2f,%.2f,%.2f,%.3f,%.4f\n" % (r + (r[5] / tot,)))
    for r in sorted(first, key=lambda r: -r[5]):
        f.write("%s,%d,%.2f,%.2f,%.2f,%.3f,\n" % r)
print(open(a.out).read())
