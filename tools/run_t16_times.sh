for shape in "1 12 8" "1 12 64" "1 12 128" "1 12 256" "1 12 384" "1 12 512" "1 12 640" "1 12 768"; do
  echo -n "B H T = $shape: "; CMHSA_T16=1 CMHSA_T16_SUB=1 CMHSA_T16_WAVES=4 timeout -k 5 60 tools/cmhsa_bench_env $shape 2000 3000 2>&1 | sed 's/plan: \([^:]*\): \([0-9]\)-wave.*most \([0-9]*\) tiles.*estimate \([0-9]*\) cycles/\1 w\2 tiles\/SIMD \3 est \4 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
done
