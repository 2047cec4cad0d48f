# mid-size and long shapes: the planner's pick (first line of a pair) against the same library with the 16-key-tile family off
for shape in "1 12 1536" "1 12 2048" "1 12 2560" "1 12 3072" "1 12 4096" "1 12 6144" "1 12 8160" "1 16 2048" "1 16 3072" "1 16 4096" "1 20 2048" "1 25 1536" "1 25 2048" "1 25 4096" \
             "2 12 1024" "2 12 1536" "2 12 2048" "2 12 4096" "3 12 1024" "3 12 3000" "4 12 1024" "4 12 2048" "2 25 1024" "2 16 2048" "5 12 1024" "6 12 1024" "3 16 1024" "4 16 1024" "2 20 1536"; do
  for mode in "CMHSA_X=0" "CMHSA_T16=0"; do
    echo -n "$shape $mode: "; env $mode timeout -k 5 60 tools/cmhsa_bench_env $shape 300 1500 2>&1 | sed 's/plan: \([^:]*\):.*estimate \([0-9]*\) cycles.*/\1 est \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done
