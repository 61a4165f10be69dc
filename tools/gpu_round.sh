#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the GPU test suite, then bench.py in the configurations DESIGN.md quotes.
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh [tag]'      results under gpurun_out/<tag>_*
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-round}
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log
grep -E "^E  |Error" gpurun_out/${tag}_tests.log | head -20
run() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > gpurun_out/${tag}_bench_$name.json 2>> gpurun_out/${tag}_bench.err
python - gpurun_out/${tag}_bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step;", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms", d["roofline"]["frac"], "of HBM;", d["valid_last_step"], "valid")
    print("   kernels", d["kernels_ms_per_context_step"])
    for k in ("parity_probe", "host_fed", "single_stream", "cpu_baseline", "pose_rmse_vs_cpu"):
        if d.get(k): print("   ", k, json.dumps(d[k])[:600])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run default
run 1ctx --contexts 1 --lanes 64 --cpu-frames 0 --host-fed-steps 0 --single-stream 0
