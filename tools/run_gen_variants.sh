# GPT2 512-token generation, HIP-graph loop, fusion variants:  bash tools/run_gen_variants.sh [model]
M=${1:-gpt2}
for extra in "" "--fused-add-ln" "--fused-add-ln --fused-step-ops"; do
  echo -n "$M graphs fused-gelu $extra: "; timeout -k 10 400 python3 llm-inference_amd/gpt2_generation.py --model $M --graphs --fused-gelu --last-position-lm-head $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['seconds'], d['tokens_generated'], d['first_tokens'][8:14])" || exit 1
done
