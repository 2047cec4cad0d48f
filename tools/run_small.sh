# attention at generation sizes (B=1, H=12, T = 64..512; GPT2-XL: H=25): us per launch, back to back, workspace plans
for H in 12 25; do for T in 64 128 192 256 320 384 448 512; do
  echo -n "H=$H T=$T: "; timeout -k 5 60 tools/cmhsa_bench 1 $H $T 2000 2000 2>&1 | sed 's/plan: \([^:]*\):.*\(SIMD cap [0-9]*\).*/\1, \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
done; done
