#!/bin/bash
# The host code's property tests (hypothesis: the layer merge and diff against their models, copy ops, the tar reader
# and writer, the header codec) with FRESH random cases: the suite runs them from a fixed seed so that a red test stays
# red; this runs MI_PROPERTY_SOAK times as many examples from a new seed each time.  No GPU needed.
#   tools/property_soak.sh [factor = 10] [asan]      (asan: against the ASan + UBSan build tools/asan_host_tests.sh makes)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FACTOR=${1:-10}
cd "$ROOT"
FILES=$(grep -l "^from hypothesis\|^import hypothesis" tests/test_*.py)
if [ "${2:-}" = asan ]; then
    MI_PROPERTY_SOAK=$FACTOR MI_ASAN_TARGET="$FILES" tools/asan_host_tests.sh
else
    MI_PROPERTY_SOAK=$FACTOR python -m pytest -q -m "not gpu" -p no:cacheprovider $FILES
fi
