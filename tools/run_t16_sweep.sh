# the planner's choice (first line of a pair) against the same library with the 16-key-tile family switched off (second):
#   bash tools/run_t16_sweep.sh
for H in 12 16 20 25; do for T in 8 40 64 100 128 200 256 320 384 448 512 640 768 1024; do
  for mode in "CMHSA_X=0" "CMHSA_T16=0"; do
    echo -n "B1 H$H T$T $mode: "; env $mode timeout -k 5 60 tools/cmhsa_bench_env 1 $H $T 1500 2000 2>&1 | sed 's/plan: \([^:]*\):.*estimate \([0-9]*\) cycles/\1 est \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done; done
for shape in "2 12 128" "2 12 512" "4 12 256" "8 12 128" "2 25 256" "4 16 512" "16 12 64" "3 12 1024"; do
  for mode in "CMHSA_X=0" "CMHSA_T16=0"; do
    echo -n "B H T $shape $mode: "; env $mode timeout -k 5 60 tools/cmhsa_bench_env $shape 1500 2000 2>&1 | sed 's/plan: \([^:]*\):.*estimate \([0-9]*\) cycles/\1 est \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done
