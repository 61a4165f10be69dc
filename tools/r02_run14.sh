#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { name=$1; shift; ( timeout 900 python bench.py --cpu-frames 0 --host-fed-steps 0 --single-stream 0 "$@" ) > gpurun_out/r02h_$name.json 2>> gpurun_out/r02h.err
python - gpurun_out/r02h_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step;", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms", d["roofline"]["frac"], "of HBM;", d["valid_last_step"], "valid")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run prio_high --det-priority high
run prio_high_2ctx --det-priority high --contexts 2 --lanes 128
run prio_low_again
run post_on_rest --post-on-rest 1
