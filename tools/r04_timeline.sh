#!/bin/bash
# GPU box: kernel trace of the benchmarked 3 x 64 schedule (8 steps), reduced to a timeline of its last steps:
# gpurun_out/<tag>_timeline.csv = queue, kernel, start_us (from the first kept launch), duration_us.  usage: gpurun -- 'bash tools/r04_timeline.sh [tag] [bench args]'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
tag=${1:-r04t}; shift
mkdir -p $O
B="--steps 6 --warmup 4 --cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --cut-steps 0 --other-workloads 0 --relief-lanes 0 --frames 176"
rm -rf /tmp/ps_tl
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_tl -- python $R/bench.py $B "$@" > $O/${tag}_timeline.log 2>&1
f=$(find /tmp/ps_tl -name "*kernel_trace.csv" | head -1)
python - "$f" $O/${tag}_timeline.csv <<'PY'
import csv, sys, re
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
PY
grep '^{' $O/${tag}_timeline.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms/step', d['ms_per_step'], 'host enqueue', d.get('host_enqueue_ms_per_step'))"
