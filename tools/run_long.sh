# single long prompts (B = 1, T = 1536..6144): the planner's choice
for H in 12 16 25; do for T in 1536 2048 3072 4096 6144; do
  echo -n "H=$H T=$T: "; timeout -k 5 60 tools/cmhsa_bench 1 $H $T 300 300 2>&1 | sed 's/plan: \([^:]*\):.*\(SIMD cap [0-9]*\).*/\1, \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
done; done
