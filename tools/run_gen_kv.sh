for extra in "" "--fused-add-ln --fused-step-ops" "--fused-add-ln --fused-step-ops --few-row-linear"; do
  echo -n "gpt2 kv-cache fused-gelu $extra: "; timeout -k 10 280 python3 llm-inference_amd/gpt2_generation.py --model gpt2 --kv-cache --fused-gelu $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['seconds'], d['tokens_generated'], d['first_tokens'][8:14])" || exit 1
done
