#!/bin/bash
# GPU box: the profile set DESIGN.md quotes -> gpurun_out/<tag>_*   (copied into profiles/ afterwards)
#   kernel stats (rocprofv3 --kernel-trace --stats; steady state = warm-up steps dropped, tools/trace_stats.py) of
#     3ctx          the benchmarked shape, 3 x 64 lanes pipelined      + the bench line of that same profiled run
#     1ctx          one context of 64 lanes alone
#     single_stream one stream alone
#   (--frames: just enough for no stream to see a frame twice in these short runs; the default renders >= 200 per trajectory)
#   PMC passes (tools/pmc_passes.py): traffic by request size, VALU / LDS counters, valu_issue_frac per kernel
# usage: bash tools/r04_prof.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
tag=${1:-r04}
mkdir -p $O
W=3; K=10
B="--steps $K --warmup $W --cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --cut-steps 0 --other-workloads 0"
run_stats() {  # name, skip, total, command...
    local name=$1 skip=$2 total=$3; shift 3
    rm -rf /tmp/ps_$name
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$name -- "$@" > $O/${tag}_prof_$name.log 2>&1
    f=$(find /tmp/ps_$name -name "*kernel_trace.csv" | head -1)
    if [ -n "$f" ]; then python $R/tools/trace_stats.py $f $O/${tag}_kernel_stats_$name.csv --skip-steps $skip --total-steps $total > /dev/null; else echo "no trace for $name"; tail -3 $O/${tag}_prof_$name.log; fi
    f=$(find /tmp/ps_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python $R/tools/summarize_prof.py $f $O/${tag}_kernel_stats_${name}_whole_run.csv > /dev/null
}
run_stats 3ctx $W $((W+K)) python $R/bench.py $B --relief-lanes 0 --frames 176
grep '^{' $O/${tag}_prof_3ctx.log | tail -1 > $O/${tag}_bench_profiled_3ctx.json
run_stats 1ctx $W $((W+K)) python $R/bench.py $B --contexts 1 --lanes 64 --relief-lanes 0 --frames 64
grep '^{' $O/${tag}_prof_1ctx.log | tail -1 > $O/${tag}_bench_profiled_1ctx.json
run_stats single_stream 1 20 python $R/tools/single_stream_bench.py --only-plain
timeout 900 python $R/tools/pmc_passes.py $tag > $O/${tag}_pmc.log 2>&1
tail -3 $O/${tag}_pmc.log
head -30 $O/${tag}_kernel_stats_3ctx.csv
python - <<PY
import json
d = json.load(open("$O/${tag}_bench_profiled_3ctx.json"))
print("profiled 3ctx run: value", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms")
PY
