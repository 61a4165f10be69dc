#!/usr/bin/env python3
"""kernel_trace.csv of the benchmarked schedule -> queue, kernel, start_us, duration_us of its last three steps (tools/prof.sh timeline)."""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_\w+", r["Kernel_Name"])
    if m:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), m.group(0)))
rows.sort()
# keep the last three steps: from the third-to-last k_begin_frame of the detect calls on (2 begin_frame launches per context-step)
begins = [i for i, x in enumerate(rows) if x[3] == "k_begin_frame"]
start = begins[-18] if len(begins) >= 18 else 0
t0 = rows[start][0]
with open(sys.argv[2], "w") as f:
    f.write("queue,kernel,start_us,duration_us\n")
    for s, e, q, k in rows[start:]:
        f.write("%s,%s,%.1f,%.1f\n" % (q, k, (s - t0) / 1e3, (e - s) / 1e3))
print("timeline rows", len(rows) - start)
