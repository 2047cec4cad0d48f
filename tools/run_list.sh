# job lists against the other families on config 4's attention shapes (2 000 warm-up launches, 300 timed):
#   bash tools/run_list.sh       CMHSA_LIST: 0 = family off, 1 = wherever it applies; CMHSA_LIST_SCHED: 1 = teams, 0 = (b,h) by (b,h)
for shape in "64 25 512" "32 25 512" "16 25 512" "64 25 256" "64 25 384" "37 25 511" "64 12 1024" "64 16 512"; do
  for mode in "CMHSA_LIST=0" "CMHSA_LIST=1 CMHSA_LIST_SCHED=1" "CMHSA_LIST=1 CMHSA_LIST_SCHED=0" "CMHSA_X=planner"; do
    echo -n "B H T = $shape $mode: "; env $mode timeout -k 5 100 tools/cmhsa_bench_env $shape 300 2000 2>&1 | sed 's/plan: \([^:]*\):\(.*\), workspace.*estimate \([0-9]*\) cycles.*/\1 \2 est \3 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done
