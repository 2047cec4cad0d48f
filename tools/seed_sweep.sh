#!/bin/bash
# different valid packings of one shape (seeded), timed: which plan features matter?  (plan-forcing build tools/ab/env)
cd "$(dirname "$0")/.."
SHAPE=$1; ST=$2; WU=$3; shift 3
for seed in "$@"; do
  out=$(CMHSA_SEED=$seed LD_LIBRARY_PATH=tools/ab/env timeout -k 5 100 tools/cmhsa_bench $SHAPE $ST $WU 2>&1 | tr '\n' ' ')
  echo "seed $seed: $(echo $out | sed 's/.*us_per_launch": \([0-9.]*\),.*/\1 us/')"
done
