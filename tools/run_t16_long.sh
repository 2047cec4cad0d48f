# single long prompts: the 16-key-tile kernel with 32-row blocks (job lists cut at tile granularity) against the 32 x 32 families
for shape in "1 12 1536" "1 12 2048" "1 12 3072" "1 12 4000" "1 16 2048" "1 16 4000" "1 25 1536" "1 25 4000" "2 12 2048"; do
  for mode in "CMHSA_T16=0" "CMHSA_T16=1 CMHSA_T16_SUB=2 CMHSA_T16_WAVES=8" "CMHSA_T16=1 CMHSA_T16_SUB=2 CMHSA_T16_WAVES=4" "CMHSA_T16=1 CMHSA_T16_SUB=1 CMHSA_T16_WAVES=8"; do
    echo -n "$shape $mode: "; env $mode timeout -k 5 60 tools/cmhsa_bench_env $shape 500 1500 2>&1 | sed 's/plan: \([^:]*\): \([0-9]\)-wave.*most \([0-9]*\) tiles.*estimate \([0-9]*\) cycles/w\2 tiles\/SIMD \3 est \4 |/; s/plan: \([^:]*\):.*estimate \([0-9]*\) cycles.*/\1 est \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done
