#!/bin/bash
# GPU box: the pipelined schedule with more / smaller contexts (rest chains in parallel).  usage: gpurun -- 'bash tools/r04_sched2.sh [tag]'
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-r04s}
mkdir -p gpurun_out
Q="--steps 40 --warmup 6 --cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --relief-lanes 0 --cut-steps 0 --other-workloads 0 --frames 210"
run() { name=$1; shift; ( timeout 400 env $ENVX python bench.py $Q "$@" ) > gpurun_out/${tag}_bench_$name.json 2>> gpurun_out/${tag}_bench.err
python - gpurun_out/${tag}_bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels_ms_per_context_step"]
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step; host enqueue", d.get("host_enqueue_ms_per_step"), "ms/step;", d["valid_last_step"], "valid; det chain %.3f rest chain %.3f" % (k.get("resize", 0) + k.get("fast", 0) + k.get("select", 0), sum(v for n, v in k.items() if n not in ("resize", "fast", "select", "begin_frame"))))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
# HWQ=n: GPU_MAX_HW_QUEUES of the HIP runtime (default 4: the streams of a process share that many hardware queues, and a
# fifth stream is serialised behind another one's kernels)
if [ -n "$HWQ" ]; then
  for q in $HWQ; do
    ENVX="GPU_MAX_HW_QUEUES=$q" run q${q}_3x64
    ENVX="GPU_MAX_HW_QUEUES=$q" run q${q}_4x48 --contexts 4 --lanes 192
    ENVX="GPU_MAX_HW_QUEUES=$q" run q${q}_4x64 --contexts 4 --lanes 256
    ENVX="GPU_MAX_HW_QUEUES=$q" run q${q}_6x32 --contexts 6 --lanes 192
  done
  exit 0
fi
run 3x64
run 4x48 --contexts 4 --lanes 192
run 4x64 --contexts 4 --lanes 256
run 6x32 --contexts 6 --lanes 192
run 5x64 --contexts 5 --lanes 320
