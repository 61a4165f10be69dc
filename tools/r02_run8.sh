#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/ubench/mfma_f64_layout > gpurun_out/mfma_f64_layout.log 2>&1; cat gpurun_out/mfma_f64_layout.log
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02_tests.log
tail -4 gpurun_out/r02_tests.log
grep -E "^E  |Error" gpurun_out/r02_tests.log | head -20
python tools/single_stream_bench.py > gpurun_out/r02_single_stream_b.json 2> gpurun_out/r02_single_stream_b.err; cat gpurun_out/r02_single_stream_b.json; tail -3 gpurun_out/r02_single_stream_b.err
( timeout 900 python bench.py --cpu-frames 0 --host-fed-steps 0 --single-stream 0 --contexts 1 --lanes 64 ) > gpurun_out/r02_bench_d_1ctx.json 2> gpurun_out/r02_bench_d.err
python - <<'PY'
import json
for f in ("r02_bench_d_1ctx",):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["valid_last_step"], d["mean_tracked"])
        print("   ", d["kernels_ms_per_context_step"])
    except Exception as e:
        print(f, "FAILED", e)
PY
