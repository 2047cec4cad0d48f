#!/bin/bash
# multi-round grids: per-launch time against the forced chain cap (plan-forcing build in tools/ab/env)
cd "$(dirname "$0")/.."
SHAPE=$1; ST=$2; WU=$3; shift 3
for c in "$@"; do
  for w in 4 8; do
    out=$(CMHSA_WAVES=$w CMHSA_CHAIN=$c CMHSA_TABLE=0 CMHSA_VERBOSE=1 LD_LIBRARY_PATH=tools/ab/env timeout -k 5 100 tools/cmhsa_bench $SHAPE $ST $WU 0 nows 2>&1 | tr '\n' ' ')
    echo "waves $w chain $c: $(echo $out | sed 's/.*wgs\/bh=\([0-9]*\).*us_per_launch": \([0-9.]*\), "tflops": \([0-9.]*\).*/\2 us \3 TF wgs\/bh=\1/')"
  done
done
