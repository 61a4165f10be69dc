#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02_tests.log
tail -4 gpurun_out/r02_tests.log
( timeout 600 python bench.py --cpu-frames 0 ) > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err
( SVO_DEBUG_MODE=12 timeout 600 python bench.py --cpu-frames 0 ) > gpurun_out/r02_bench_b_nospec.json 2>> gpurun_out/r02_bench_b.err
( timeout 600 python bench.py --cpu-frames 0 --contexts 1 --lanes 64 ) > gpurun_out/r02_bench_b_1ctx.json 2>> gpurun_out/r02_bench_b.err
python - <<'PY'
import json
for f in ("r02_bench_b", "r02_bench_b_nospec", "r02_bench_b_1ctx"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["valid_last_step"], d["mean_tracked"])
        print("   ", d["kernels_ms_per_context_step"])
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 gpurun_out/r02_bench_b.err
