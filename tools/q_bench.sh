cd ${GRAFT_REPO_ROOT:-.}
Q="--cpu-frames 0 --host-fed-steps 0 --single-stream 0"
run() { echo "== $*"; timeout 600 python bench.py $Q "$@" 2>gpurun_out/q_err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('exclusive'), d['kernels_ms_per_context_step'])" || tail -3 gpurun_out/q_err.log; }
run "$@"
