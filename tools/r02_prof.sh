#!/bin/bash
# GPU box: rocprofv3 kernel stats (3 contexts, 1 context, single stream) + PMC traffic + VALU/LDS counters -> gpurun_out/
# (the summaries DESIGN.md quotes are copied from there into profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
B="--steps 10 --warmup 3 --cpu-frames 0 --host-fed-steps 0 --single-stream 0"
run_stats() {  # name, command...
    local name=$1; shift
    rm -rf /tmp/ps_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$name -- "$@" > /tmp/ps_$name.log 2>&1
    f=$(find /tmp/ps_$name -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then python $R/tools/summarize_prof.py $f $O/r02_kernel_stats_$name.csv > /dev/null; else echo "no stats for $name"; tail -3 /tmp/ps_$name.log; fi
}
run_stats 3ctx python $R/bench.py $B
run_stats 1ctx python $R/bench.py $B --contexts 1 --lanes 64
run_stats single_stream python $R/tools/single_stream_bench.py
# (round 2 ran tools/pmc_traffic.py here; that script was folded into tools/pmc_passes.py in round 3)
python $R/tools/pmc_passes.py r02 > $O/r02_pmc_traffic.log 2>&1
for i in 1 2; do
  if [ $i = 1 ]; then set="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; else set="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; fi
  rm -rf /tmp/pv$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pv$i -- python $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --host-fed-steps 0 --single-stream 0 --contexts 1 --lanes 64 > /tmp/pv$i.log 2>&1
  f=$(find /tmp/pv$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f > $O/r02_valu_lds_$i.csv; else echo "no counters pass $i"; tail -3 /tmp/pv$i.log; fi
done
cd $R
for wl in config3 config5; do
  ( timeout 600 python bench.py --workload $wl --cpu-frames 0 --host-fed-steps 0 --single-stream 0 ) > $O/r02_bench_$wl.json 2>> $O/r02_bench_other.err
done
( timeout 900 python bench.py ) > $O/r02_bench_final.json 2> $O/r02_bench_final.err
python - <<'PY'
import json
for f in ("r02_bench_config3", "r02_bench_config5", "r02_bench_final"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        print(f, d["metric"], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("exclusive"), d["valid_last_step"], d["mean_kps"], d["mean_tracked"])
    except Exception as e:
        print(f, "FAILED", e)
PY
head -24 $O/r02_kernel_stats_1ctx.csv
