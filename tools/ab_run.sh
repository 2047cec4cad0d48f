#!/bin/bash
# A/B timing of tools/ab/*/libcmhsa_hip.so on one shape, variants interleaved, ROUNDS passes:
#   tools/ab_run.sh "B H T steps warmup" ROUNDS name1 name2 ...
cd "$(dirname "$0")/.."
ARGS=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    out=$(LD_LIBRARY_PATH=tools/ab/$v timeout -k 5 60 tools/cmhsa_bench $ARGS 2>/dev/null) || { echo "$v FAILED"; exit 1; }
    echo "$v $(echo $out | sed 's/.*us_per_launch": \([0-9.]*\).*o00": \([0-9.-]*\).*/\1 us  o00=\2/')"
  done
done
