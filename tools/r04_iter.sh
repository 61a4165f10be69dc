#!/bin/bash
# GPU box: one build -> measure iteration: the detector / matcher parity tests first (fail fast), a short bench line, the 1-context
# kernel stats and (with PMC=1) the counter passes.  PARITY_ENV="NAME=VALUE": the parity subset runs under that environment.  usage: gpurun -- 'PMC=1 bash tools/r04_iter.sh <tag> [extra bench args]'
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-r04i}; shift
mkdir -p gpurun_out
( time timeout 900 env $PARITY_ENV python -m pytest tests/test_gpu_parity.py -x -q -k "small_sequence or pyramid_and_detector or noise_images or full_size or hamming_match or speculative_fast or fast_orb_multi or random_parameter_sets_match or sixty_four" ) > gpurun_out/${tag}_first.log 2>&1
echo "first rc=$?" >> gpurun_out/${tag}_first.log
tail -6 gpurun_out/${tag}_first.log
Q="--steps 40 --warmup 6 --cpu-frames 12 --long-steps 0 --host-fed-steps 0 --single-stream 0 --relief-lanes 0 --cut-steps 0 --other-workloads 0 --frames 210"
( time timeout 600 python bench.py $Q "$@" ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python - gpurun_out/${tag}_bench.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "pairs/s", d["ms_per_step"], "ms/step;", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], "ms;", d["valid_last_step"], "valid; kps", d["mean_kps"], "matches", d["mean_matches"], "tracked", d["mean_tracked"])
    print("   exclusive", json.dumps(d["roofline"].get("exclusive")))
    print("   kernels", d["kernels_ms_per_context_step"])
    print("   parity", json.dumps(d["parity_probe"])[:400])
    print("   legs", d.get("legs_s"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
tail -4 gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="--steps 10 --warmup 3 --cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --cut-steps 0 --other-workloads 0 --relief-lanes 0"
rm -rf /tmp/ps_1ctx
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_1ctx -- python $R/bench.py $B --contexts 1 --lanes 64 --frames 64 > $O/${tag}_prof_1ctx.log 2>&1
f=$(find /tmp/ps_1ctx -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python $R/tools/trace_stats.py $f $O/${tag}_kernel_stats_1ctx.csv --skip-steps 3 --total-steps 13 > /dev/null
head -19 $O/${tag}_kernel_stats_1ctx.csv
# A/B legs: AB="NAME=VALUE NAME2=VALUE2 ..." -> one more 1-context kernel-stats pass per assignment, the changed rows printed
for ab in $AB; do
  rm -rf /tmp/ps_ab
  env $ab timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_ab -- python $R/bench.py $B --contexts 1 --lanes 64 --frames 64 > $O/${tag}_prof_ab_${ab}.log 2>&1
  f=$(find /tmp/ps_ab -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $R/tools/trace_stats.py $f $O/${tag}_kernel_stats_1ctx_${ab}.csv --skip-steps 3 --total-steps 13 > /dev/null
  echo "A/B $ab:"; head -8 $O/${tag}_kernel_stats_1ctx_${ab}.csv
  grep '^{' $O/${tag}_prof_ab_${ab}.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value', d['value'], 'valid', d['valid_last_step'], 'tracked', d['mean_tracked'])"
done
if [ -n "$PMC" ]; then
  ( time timeout 1200 python $R/tools/pmc_passes.py $tag ) > $O/${tag}_pmc.log 2>&1
  tail -4 $O/${tag}_pmc.log
  python - $O/${tag}_pmc.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    tot = 0
    tot = 0.0
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["counters"].get("SQ_INSTS_VALU", 0) * kv[1].get("launches_per_step", 1)):
        c = v["counters"]; n = v.get("launches_per_step", 1); tot += c.get("SQ_INSTS_VALU", 0) * n
        print("%-24s valu %8.2fM x%.1f  /wave %7.1f  issue %.2f  lds_conf %.2f  mfma %7.2fM  cycles %8d  rd %.1f MB wr %.1f MB" % (k, c.get("SQ_INSTS_VALU", 0) / 1e6, n, v.get("valu_per_wave", 0), v.get("valu_issue_frac", 0), v.get("lds_conflict_frac", 0), c.get("SQ_INSTS_MFMA", 0) / 1e6, v.get("cycles", 0), v.get("read_bytes", 0) / 1e6, v.get("write_bytes", 0) / 1e6))
    print("total VALU wave-instructions per context-step: %.1f M; path %.2f MB/pair" % (tot / 1e6, d.get("path_MB_per_pair", 0)))
except Exception as e:
    print("pmc summary failed", e)
PY
fi
