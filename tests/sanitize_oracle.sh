#!/bin/bash
# The CPU suite against the ASan + UBSan build of the C oracle (sanitizer run of the native code that
# runs on the host).  Any out-of-bounds access, use-after-free, signed overflow, misaligned access or invalid shift in
# oracle/rrl_oracle.c aborts the test that triggers it.
#   bash tests/sanitize_oracle.sh [pytest args]
set -eu
cd "$(dirname "$0")/.."
make -C oracle -s -B sanitize
export RRL_ORACLE_SANITIZE=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
exec python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
