#!/bin/bash
# GPU box: a subset of the parity tests first (fast feedback), then the whole GPU suite, then a short bench.  usage: gpurun -- 'bash tools/r04_quick.sh [tag] [pytest -k expr]'
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-r04q}
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_sequence or pyramid_and_detector or ransac_kernel_form or full_size" ) > gpurun_out/${tag}_first.log 2>&1
echo "first rc=$?" >> gpurun_out/${tag}_first.log
tail -25 gpurun_out/${tag}_first.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -30 gpurun_out/${tag}_tests.log
( timeout 600 python bench.py --steps 40 --warmup 6 --cpu-frames 12 --long-steps 0 --host-fed-steps 0 --single-stream 0 --relief-lanes 0 ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - gpurun_out/${tag}_bench.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step;", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms;", d["valid_last_step"], "valid; kps", d["mean_kps"], "matches", d["mean_matches"], "tracked", d["mean_tracked"])
    print("   kernels", d["kernels_ms_per_context_step"])
    print("   funnel", d["track_funnel_mean"])
    print("   parity", json.dumps(d["parity_probe"])[:500])
    print("   pose", json.dumps(d["pose_rmse_vs_cpu"])[:600])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
tail -5 gpurun_out/${tag}_bench.err
