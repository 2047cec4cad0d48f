#!/bin/bash
# rocprofv3 kernel statistics of one GPT2 generation run (HIP-graph loop + fused GELU):  bash tools/prof_gen.sh [model]
MODEL=${1:-gpt2}
EXTRA=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_gen_$MODEL
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/llm-inference_amd/gpt2_generation.py" --model $MODEL --graphs --fused-gelu --last-position-lm-head $EXTRA > "$OUT/run.log" 2>&1 || { echo FAILED; tail -5 "$OUT/run.log"; exit 1; }
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
tail -3 "$OUT/run.log"
