#!/bin/bash
# build everything, then run a command on the MI355X box: tools/gpu.sh <timeout-s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|Error" && exit 1
T=${1:-600}; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
