#!/bin/bash
# Run ON THE GPU BOX: L2->fabric read requests by size (exact bytes = 32*n32 + 64*n64 + 128*n128), per kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 3 --warmup 1 --cpu-frames 0"
rm -rf /tmp/pr
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/pr -- $CMD > /tmp/pr.log 2>&1
f=$(find /tmp/pr -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then tail -5 /tmp/pr.log; else python $R/tools/pmc_summary.py $f; fi
