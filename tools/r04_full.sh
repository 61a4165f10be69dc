#!/bin/bash
# GPU box: the whole GPU suite, the default bench line, then the profile set.  usage: gpurun -- 'bash tools/r04_full.sh [tag]'
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-r04}
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/${tag}_tests.log
grep -E "^FAILED|passed|failed" gpurun_out/${tag}_tests.log | tail -12
( time timeout 1200 python bench.py ) > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python - gpurun_out/${tag}_bench_default.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step;", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms;", d["valid_last_step"], "valid; kps", d["mean_kps"], "matches", d["mean_matches"], "tracked", d["mean_tracked"], "redo", d["fast_redo_rate"], "timed", d["timed_region_s"], "render", d["config"]["render_s"])
    print("   roofline", json.dumps(d["roofline"].get("exclusive")), d["roofline"]["frac"])
    print("   kernels", d["kernels_ms_per_context_step"])
    print("   funnel", d["track_funnel_mean"]); print("   legs", d.get("legs_s"), "host enqueue", d.get("host_enqueue_ms_per_step"))
    for k in ("parity_probe", "scene_cuts", "other_workloads", "host_fed", "single_stream", "cpu_baseline", "pose_rmse_vs_cpu", "other_scene"):
        print("   ", k, json.dumps(d.get(k))[:900])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
tail -3 gpurun_out/${tag}_bench_default.err
bash tools/r04_prof.sh $tag 2>&1 | tail -45
