#!/bin/bash
# Run ON THE GPU BOX (via gpurun): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (never combined with
# sys/runtime traces), per-kernel averages written to gpurun_out/pmc_fetch.csv and gpurun_out/pmc_write.csv.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 $PMC_EXTRA"
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $CMD > /dev/null 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pw -name "*counter_collection.csv" | head -1) > $R/gpurun_out/pmc_write.csv
