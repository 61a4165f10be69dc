#!/bin/bash
# GPU box: engine clock and power while the benchmarked schedule runs (is the step short of its issue bound because the part throttles?).
# usage: gpurun -- 'bash tools/r04_clocks.sh [tag]'
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-r04clk}; shift
mkdir -p gpurun_out
Q="--steps 1500 --warmup 6 --cpu-frames 0 --long-steps 0 --host-fed-steps 0 --single-stream 0 --exclusive 0 --relief-lanes 0 --cut-steps 0 --other-workloads 0 --frames 210"
( python bench.py $Q "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err ) &
pid=$!
: > gpurun_out/${tag}_smi.log
while kill -0 $pid 2>/dev/null; do
  echo "t $(date +%s.%N)" >> gpurun_out/${tag}_smi.log
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> gpurun_out/${tag}_smi.log
  sleep 0.2
done
wait $pid
python - gpurun_out/${tag}_bench.json gpurun_out/${tag}_smi.log <<'PY'
import json, sys, re
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "timed", d["timed_region_s"], "s")
s = open(sys.argv[2]).read()
sc = [int(x) for x in re.findall(r"sclk clock level: \w+: \((\d+)Mhz\)", s)]
pw = [float(x) for x in re.findall(r"Power \(W\): ([\d.]+)", s)]
print("sclk samples", len(sc), "max", max(sc) if sc else None, "last 12", sc[-12:])
print("power samples", len(pw), "max", max(pw) if pw else None, "last 12", pw[-12:])
PY
tail -30 gpurun_out/${tag}_smi.log | head -12
