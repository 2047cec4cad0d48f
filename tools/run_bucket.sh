# HIP-graph generation loop: bucket size sweep (GPT2-small, 512 tokens)
for b in 64 32 16 8; do
  echo -n "bucket $b: "; timeout -k 10 280 python3 llm-inference_amd/gpt2_generation.py --model gpt2 --graphs --fused-gelu --last-position-lm-head --bucket $b 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['seconds'], d['warmup_seconds'], d['tokens_generated'], d['first_tokens'][8:14])" || exit 1
done
