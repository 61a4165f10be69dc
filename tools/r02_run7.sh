#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02_tests.log
tail -4 gpurun_out/r02_tests.log
grep -E "^E  |Error" gpurun_out/r02_tests.log | head -20
