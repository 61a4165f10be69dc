#!/bin/bash
# GPU box: the schedule variants of round 3 re-measured with the round-4 kernels.  usage: gpurun -- 'bash tools/r04_sched3.sh [tag]'
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-r04s3}
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "small_sequence or pyramid_and_detector or full_size or fast_orb_multi or sixty_four" ) > gpurun_out/${tag}_first.log 2>&1
tail -3 gpurun_out/${tag}_first.log
Q="--steps 40 --warmup 6 --cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --relief-lanes 0 --cut-steps 0 --other-workloads 0 --frames 210"
run() { name=$1; shift; ( timeout 400 env $ENVX python bench.py $Q "$@" ) > gpurun_out/${tag}_bench_$name.json 2>> gpurun_out/${tag}_bench.err
python - gpurun_out/${tag}_bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels_ms_per_context_step"]
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step; host enqueue", d.get("host_enqueue_ms_per_step"), "ms/step;", d["valid_last_step"], "valid")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run default
run select_on_rest --post-on-rest 3
run det2 --det-streams 2
run detlow --det-priority low
run rest2 --rest-streams 2
run noahead --detect-ahead 0
run post0 --post-on-rest 0
