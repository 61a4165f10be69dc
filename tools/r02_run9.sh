#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/single_stream_bench.py > gpurun_out/r02_single_stream_c.json 2> /dev/null; cat gpurun_out/r02_single_stream_c.json
for cfg in "2 128" "3 192" "4 256" "4 192" "6 192"; do
  set -- $cfg
  ( timeout 900 python bench.py --cpu-frames 0 --host-fed-steps 0 --single-stream 0 --contexts $1 --lanes $2 ) > gpurun_out/r02_tune_$1_$2.json 2>> gpurun_out/r02_tune.err
  python - $1 $2 <<'PY'
import json, sys
f = "gpurun_out/r02_tune_%s_%s.json" % (sys.argv[1], sys.argv[2])
try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print("contexts", sys.argv[1], "lanes", sys.argv[2], "->", d["value"], "pairs/s", d["ms_per_step"], "ms/step; fast", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print(f, "FAILED", e)
PY
done
