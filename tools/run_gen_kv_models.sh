# opt-in KV-cache loop, 512 tokens, every model size: stock linear layers vs the few-row HIP linear kernels
for m in gpt2 gpt2-medium gpt2-large gpt2-xl; do
  for extra in "" "--few-row-linear"; do
    echo -n "$m kv-cache all fusions $extra: "; timeout -k 10 280 python3 llm-inference_amd/gpt2_generation.py --model $m --kv-cache --fused-gelu --fused-add-ln --fused-step-ops $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['seconds'], d['tokens_generated'], d['first_tokens'][8:14])" || exit 1
  done
done
