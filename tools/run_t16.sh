# 16-row-block family against the planner's other choice, B=1 (generation sizes) and a few more:  bash tools/run_t16.sh
for shape in "1 12 64" "1 12 128" "1 12 256" "1 12 384" "1 12 512" "1 12 768" "1 12 1024" "1 25 256" "1 25 512" "1 25 1024" "1 16 1024" "1 20 1024" "2 12 1024" "1 12 2048"; do
  for t16 in 0 1; do
    echo -n "B H T = $shape T16=$t16: "; CMHSA_T16=$t16 timeout -k 5 60 tools/cmhsa_bench_env $shape 2000 3000 2>&1 | sed 's/plan: \([^:]*\):.*estimate \([0-9]*\) cycles/\1 est \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done
