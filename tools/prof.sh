#!/bin/bash
# Run ON THE GPU BOX (through gpurun).  ONE script for every build -> measure step of a round; results under gpurun_out/<tag>_*
# (copy what is to be judged into profiles/).
#
#   gpurun --timeout T -- 'bash tools/prof.sh <tag> <step> [<step> ...]'          extra bench.py arguments: BENCH_ARGS="..."
#
# steps
#   first     the fail-fast parity subset (detector, matcher, RANSAC forms, full-size streams, parameter fuzz); stops the script on a failure
#             PARITY_ENV="NAME=VALUE" runs it under that environment, PARITY_K="expr" replaces the -k expression
#   tests     the whole GPU suite (pytest -m gpu)
#   bench     a short bench line (40 timed steps, 12-frame parity probe, no side legs)
#   default   bench.py as the driver calls it (--gpus 1 --steps 20 --warmup 5: every leg, several minutes)
#   stats3    rocprofv3 --kernel-trace --stats of the benchmarked shape (2 x 96 lanes since round 6; the files keep the `3ctx` tag of rounds 2-5), steady state
#   stats1    the same with one context of 64 lanes alone (exclusive kernel times)
#   stats_ss  one stream alone (tools/single_stream_bench.py)
#   pmc       the counter passes (tools/pmc_passes.py): traffic by request size, VALU / LDS / MFMA counters, issue fractions
#   ab        AB="NAME=VALUE ..." : one more stats1 pass per assignment
#   abbench   AB="NAME=VALUE ..." : the short un-profiled bench line (60 timed steps, no CPU legs) once plainly and once per assignment
#   clocks    rocm-smi engine clock / power sampled while a 1500-step run is in flight
#   timeline  kernel timeline of 6 steady-state steps of the benchmarked shape (per-stream co-execution)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
tag=${1:-prof}; shift
mkdir -p $O
SIDE="--cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --cut-steps 0 --other-workloads 0 --relief-lanes 0"
PB="--steps 10 --warmup 3 $SIDE"

line_summary() {   # bench json -> a few lines on stdout
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(sys.argv[1].split("/")[-1], d["value"], "pairs/s", d["ms_per_step"], "ms/step;", r["kernel"], r["avg_launch_ms"], "ms", r["frac"], "of HBM;", d["valid_last_step"], "valid; kps", d["mean_kps"], "matches", d["mean_matches"], "tracked", d["mean_tracked"])
    print("   exclusive", json.dumps(r.get("exclusive")))
    print("   kernels", d["kernels_ms_per_context_step"])
    for k in ("parity_probe", "long_run", "scene_cuts", "host_fed", "single_stream", "cpu_baseline", "pose_rmse_vs_cpu", "clocks", "legs_s"):
        if d.get(k): print("   ", k, json.dumps(d[k])[:700])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}

stats() {  # name, skip, total, command...
    local name=$1 skip=$2 total=$3; shift 3
    rm -rf /tmp/ps_$name
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$name -- "$@" > $O/${tag}_prof_$name.log 2>&1
    f=$(find /tmp/ps_$name -name "*kernel_trace.csv" | head -1)
    if [ -n "$f" ]; then python $R/tools/trace_stats.py $f $O/${tag}_kernel_stats_$name.csv --skip-steps $skip --total-steps $total > /dev/null; else echo "no trace for $name"; tail -3 $O/${tag}_prof_$name.log; fi
    f=$(find /tmp/ps_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python $R/tools/summarize_prof.py $f $O/${tag}_kernel_stats_${name}_whole_run.csv > /dev/null
    grep '^{' $O/${tag}_prof_$name.log | tail -1 > $O/${tag}_bench_profiled_$name.json
}

for step in "$@"; do
  echo "=== $step"
  case $step in
  first)
    K=${PARITY_K:-"small_sequence or pyramid_and_detector or noise_images or full_size or hamming_match or speculative_fast or fast_orb_multi or random_parameter_sets_match or sixty_four or ransac_kernel_form or change_in_pose or five_thousand"}
    ( cd $R; time timeout 900 env $PARITY_ENV python -m pytest tests/test_gpu_parity.py -x -q -k "$K" ) > $O/${tag}_first.log 2>&1
    rc=$?; echo "first rc=$rc" >> $O/${tag}_first.log; tail -6 $O/${tag}_first.log
    if [ $rc -ne 0 ]; then grep -E "^E  |Error|assert" $O/${tag}_first.log | head -30; exit 1; fi ;;
  tests)
    ( cd $R; time timeout 1800 python -m pytest tests -m gpu -q ) > $O/${tag}_tests.log 2>&1
    echo "tests rc=$?" >> $O/${tag}_tests.log; tail -8 $O/${tag}_tests.log; grep -E "^E  |^FAILED" $O/${tag}_tests.log | head -30 ;;
  bench)
    ( cd $R; time timeout 600 python bench.py --steps 40 --warmup 6 --cpu-frames 12 --long-steps 0 --host-fed-steps 0 --single-stream 0 --relief-lanes 0 --cut-steps 0 --other-workloads 0 --frames 210 $BENCH_ARGS ) > $O/${tag}_bench.json 2> $O/${tag}_bench.err
    line_summary $O/${tag}_bench.json; tail -3 $O/${tag}_bench.err ;;
  default)
    ( cd $R; time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 $BENCH_ARGS ) > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
    line_summary $O/${tag}_bench_default.json; tail -3 $O/${tag}_bench_default.err ;;
  stats3)
    stats 3ctx 3 13 python $R/bench.py $PB --frames 176 $BENCH_ARGS
    head -24 $O/${tag}_kernel_stats_3ctx.csv; line_summary $O/${tag}_bench_profiled_3ctx.json | head -1 ;;
  stats1)
    stats 1ctx 3 13 python $R/bench.py $PB --contexts 1 --lanes 64 --frames 64 $BENCH_ARGS
    head -24 $O/${tag}_kernel_stats_1ctx.csv ;;
  stats_ss)
    stats single_stream 1 20 python $R/tools/single_stream_bench.py --only-plain
    head -36 $O/${tag}_kernel_stats_single_stream.csv ;;
  ab)
    for ab in $AB; do
      rm -rf /tmp/ps_ab
      env $ab timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_ab -- python $R/bench.py $PB --contexts 1 --lanes 64 --frames 64 $BENCH_ARGS > $O/${tag}_prof_ab_${ab}.log 2>&1
      f=$(find /tmp/ps_ab -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python $R/tools/trace_stats.py $f $O/${tag}_kernel_stats_1ctx_${ab}.csv --skip-steps 3 --total-steps 13 > /dev/null
      echo "A/B $ab:"; head -10 $O/${tag}_kernel_stats_1ctx_${ab}.csv
      grep '^{' $O/${tag}_prof_ab_${ab}.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value', d['value'], 'valid', d['valid_last_step'], 'tracked', d['mean_tracked'])"
    done ;;
  abbench)
    for ab in BASE=1 $AB; do
      nm=$(echo "$ab" | sed 's|.*/||' | tr '=,' '__')
      ( cd $R; env $(echo $ab | tr ',' ' ') timeout 600 python bench.py --steps 60 --warmup 6 $SIDE --frames 210 $BENCH_ARGS ) > $O/${tag}_abbench_${nm}.json 2> $O/${tag}_abbench_${nm}.err
      python - $O/${tag}_abbench_${nm}.json "$nm" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels_ms_per_context_step"]
    print("%-22s %9.1f pairs/s  %.4f ms/step  valid %s tracked %.1f | fast %.3f gn %.3f nms %.3f ransac_cnt %.3f track_filter %.3f hamming %.3f" % (sys.argv[2], d["value"], d["ms_per_step"], d["valid_last_step"], d["mean_tracked"],
          k.get("fast", 0), k.get("gauss_newton", 0), k.get("nms_rowsort", 0), sum(v for n, v in k.items() if n.startswith("ransac_cnt")), k.get("track_filter", 0), sum(v for n, v in k.items() if n.startswith("hamming"))))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    done ;;
  pmc)
    ( time timeout 1500 python $R/tools/pmc_passes.py $tag $BENCH_ARGS ) > $O/${tag}_pmc.log 2>&1
    tail -3 $O/${tag}_pmc.log
    python $R/tools/pmc_table.py $O/${tag}_pmc.json ;;
  clocks)
    ( cd $R; python bench.py --steps 1500 --warmup 6 $SIDE --frames 210 $BENCH_ARGS > $O/${tag}_clk_bench.json 2> $O/${tag}_clk_bench.err ) &
    pid=$!
    : > $O/${tag}_smi.log
    while kill -0 $pid 2>/dev/null; do
      echo "t $(date +%s.%N)" >> $O/${tag}_smi.log
      rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $O/${tag}_smi.log
      sleep 0.2
    done
    wait $pid
    line_summary $O/${tag}_clk_bench.json | head -1
    python - $O/${tag}_smi.log <<'PY'
import re, sys
s = open(sys.argv[1]).read()
sc = [int(x) for x in re.findall(r"sclk clock level: \w+: \((\d+)Mhz\)", s)]
pw = [float(x) for x in re.findall(r"Power \(W\): ([\d.]+)", s)]
print("sclk samples", len(sc), "max", max(sc) if sc else None, "last 12", sc[-12:])
print("power samples", len(pw), "max", max(pw) if pw else None, "last 12", pw[-12:])
PY
    ;;
  timeline)
    rm -rf /tmp/ps_tl
    timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps_tl -- python $R/bench.py $PB --frames 176 $BENCH_ARGS > $O/${tag}_prof_tl.log 2>&1
    f=$(find /tmp/ps_tl -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python $R/tools/timeline.py $f $O/${tag}_timeline.csv ;;
  *) echo "unknown step $step" ;;
  esac
done
