# 16-key-tile family against the planner's other choice, B=1 (generation sizes) and a few more:  bash tools/run_t16.sh
# (T16=0: never; sub=1 / sub=2: forced on with 16- / 32-row blocks)
for shape in "1 12 64" "1 12 128" "1 12 256" "1 12 384" "1 12 512" "1 12 768" "1 12 1024" "1 25 512" "1 25 1024" "1 16 1024" "2 12 1024" "1 12 2048"; do
  for mode in "CMHSA_T16=0" "CMHSA_T16=1 CMHSA_T16_SUB=1" "CMHSA_T16=1 CMHSA_T16_SUB=2"; do
    echo -n "B H T = $shape $mode: "; env $mode timeout -k 5 60 tools/cmhsa_bench_env $shape 2000 3000 2>&1 | sed 's/plan: \([^:]*\): \([0-9]\)-wave.*most \([0-9]*\) tiles.*estimate \([0-9]*\) cycles/\1 w\2 tiles\/SIMD \3 est \4 |/; s/plan: \([^:]*\):.*estimate \([0-9]*\) cycles/\1 est \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done
