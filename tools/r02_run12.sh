#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02_tests.log
tail -4 gpurun_out/r02_tests.log
grep -E "^E  |Error" gpurun_out/r02_tests.log | head -20
run() { name=$1; shift; ( timeout 900 python bench.py --cpu-frames 0 --host-fed-steps 0 --single-stream 0 "$@" ) > gpurun_out/r02_bench_$name.json 2>> gpurun_out/r02_bench_f.err
python - $name <<'PY'
import json, sys
f = "gpurun_out/r02_bench_%s.json" % sys.argv[1]
try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step; fast", d["roofline"]["avg_launch_ms"], d["valid_last_step"], d["mean_tracked"])
    print("   ", d["kernels_ms_per_context_step"])
except Exception as e:
    print(f, "FAILED", e)
PY
}
run f_1ctx --contexts 1 --lanes 64
SVO_DEBUG_MODE=14 run f_1ctx_valu --contexts 1 --lanes 64
run f_3ctx
