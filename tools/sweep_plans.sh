#!/bin/bash
# Plan sweep for tools/cmhsa_bench: the model's choice vs forced (W, c, table) combinations, same run.
# usage: tools/sweep_plans.sh "B H T" ["B H T" ...]
cd "$(dirname "$0")"
for cfg in "$@"; do
  set -- $cfg; T=$3
  echo "== $cfg (model): $(CMHSA_VERBOSE=1 timeout -k 10 60 ./cmhsa_bench $cfg 30 5 2>&1 | grep -E 'plan|us_per' | sort -u | sed -e 's/.*-> //' -e 's/.*us_per_launch": \([0-9.]*\).*/us=\1/' | tr '\n' ' ')"
  for W in 8 4; do
    cmin=$(( ((T + 31) / 32 + W - 1) / W ))
    for c in $cmin $(( cmin * 3 / 2 )) $(( cmin * 2 )); do
      for tb in 0 1; do
        r=$(CMHSA_WAVES=$W CMHSA_CHAIN=$c CMHSA_TABLE=$tb timeout -k 10 60 ./cmhsa_bench $cfg 30 5 2>&1 | grep us_per | sed -e 's/.*us_per_launch": \([0-9.]*\).*/\1/')
        echo "   W=$W c=$c table=$tb : $r us"
      done
    done
  done
done
