#!/bin/bash
# GPU box: full GPU test suite, then the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02_tests.log
( time timeout 600 python bench.py ) > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err
tail -3 gpurun_out/r02_tests.log; cat gpurun_out/r02_bench_a.json | head -c 6000; tail -5 gpurun_out/r02_bench_a.err
