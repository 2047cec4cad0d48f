set -e
timeout -k 10 400 python -u -m pytest tests/test_attention_gpu.py -x -q -k "forced_plans or one_wave or config5 or config4" --timeout 300 --timeout-method thread > gpurun_out/w1_tests.log 2>&1
echo tests-ok
for shape in "32 2 4096 40 20" "64 25 512 40 20" "32 12 1024 60 30" "16 16 512 200 100" "4 16 2048 100 50" "1 12 1024 3000 3000"; do
  echo "== $shape"
  for e in "X=0" "X=1"; do
  env $e LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*estimate/  \1 est/; s/"steps.*us_per_launch/us/; s/"frac.*//' | paste - - || exit 1
  done
done
