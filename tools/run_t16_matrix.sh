# every shape of the 16-key-tile kernel against the 32 x 32 families, B1 H12:   bash tools/run_t16_matrix.sh
for T in 384 512 640 768 1024; do
  for mode in "CMHSA_T16=0" "CMHSA_T16=1 CMHSA_T16_SUB=1 CMHSA_T16_WAVES=4" "CMHSA_T16=1 CMHSA_T16_SUB=1 CMHSA_T16_WAVES=8" "CMHSA_T16=1 CMHSA_T16_SUB=2 CMHSA_T16_WAVES=4" "CMHSA_T16=1 CMHSA_T16_SUB=2 CMHSA_T16_WAVES=8"; do
    echo -n "T$T $mode: "; env $mode timeout -k 5 60 tools/cmhsa_bench_env 1 12 $T 3000 3000 2>&1 | sed 's/plan: \([^:]*\): \([0-9]\)-wave.*most \([0-9]*\) tiles.*estimate \([0-9]*\) cycles/w\2 tiles\/SIMD \3 est \4 |/; s/plan: \([^:]*\):.*estimate \([0-9]*\) cycles.*/\1 est \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
  done
done
