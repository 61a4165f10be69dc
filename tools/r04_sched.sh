#!/bin/bash
# GPU box: the round-4 schedule (detector ahead of the previous frame's stages 3-5): its tests, then the bench under the schedules
# DESIGN.md compares.  usage: gpurun -- 'bash tools/r04_sched.sh [tag]'
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-r04s}
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_batch_host.py -x -q ) > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -6 gpurun_out/${tag}_tests.log
Q="--steps 40 --warmup 6 --cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --relief-lanes 0"
run() { name=$1; shift; ( timeout 600 python bench.py $Q "$@" ) > gpurun_out/${tag}_bench_$name.json 2>> gpurun_out/${tag}_bench.err
python - gpurun_out/${tag}_bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step;", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms;", d["valid_last_step"], "valid; tracked", d["mean_tracked"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run noahead_3ctx --detect-ahead 0
run ahead_3ctx
run ahead_4ctx --contexts 4 --lanes 256
run ahead_5ctx --contexts 5 --lanes 320
run ahead_4x48 --contexts 4 --lanes 192
run ahead_3ctx_2det --det-streams 2
run ahead_4ctx_2det --contexts 4 --lanes 256 --det-streams 2
run ahead_3ctx_detlow --det-priority low
run ahead_3ctx_select --post-on-rest 3
run ahead_6x32 --contexts 6 --lanes 192
