set -e
timeout -k 10 500 python -u -m pytest tests/test_workspace_gpu.py tests/test_lds_poison.py -x -q --timeout 300 --timeout-method thread > gpurun_out/seg8_tests.log 2>&1
echo tests-ok
bash tools/run_long.sh
for shape in "1 12 1024 3000 3000" "3 12 1024 300 300" "4 12 1024 300 300" "2 12 2048 200 200"; do echo -n "$shape: "; timeout -k 5 60 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*\(SIMD cap [0-9]*\).*/\1, \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - -; done
