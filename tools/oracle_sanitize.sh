#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer: builds oracle/_san/libsvo_oracle.so and runs every CPU test that
# drives the oracle on it (golden vectors, unit tests, the independent readings, the third-party vectors).  Test infrastructure only.
# usage: bash tools/oracle_sanitize.sh [extra pytest arguments]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$R/oracle" san
ASAN=$(gcc -print-file-name=libasan.so)
cd "$R"
LD_PRELOAD="$ASAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  SVO_ORACLE_SO="$R/oracle/_san/libsvo_oracle.so" \
  python -m pytest -x -q -m "not gpu" tests/test_oracle_golden.py tests/test_oracle_units.py tests/test_independent_reading.py \
      tests/test_independent_own_logic.py tests/test_oracle_thirdparty.py "$@"
