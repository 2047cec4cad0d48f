set -e
timeout -k 10 600 python -u -m pytest tests/test_layernorm.py tests/test_generation.py tests/test_abi.py -x -q --timeout 400 --timeout-method thread > gpurun_out/ln_tests.log 2>&1
echo tests-ok
for extra in "" "--fused-add-ln"; do
  echo -n "gpt2 graphs fused-gelu $extra: "; timeout -k 10 280 python3 llm-inference_amd/gpt2_generation.py --model gpt2 --graphs --fused-gelu --last-position-lm-head $extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['seconds'], d['tokens_generated'], d['first_tokens'][8:14])"
done
