# the same shapes after a short and after a long warm-up (shader clock transient):  bash tools/run_warm.sh
for shape in "16 16 512" "64 25 512" "16 12 1024" "4 12 1024" "32 2 4096"; do
  for sw in "100 50" "400 2000" "400 6000"; do
    echo -n "$shape steps/warmup $sw: "; timeout -k 5 100 tools/cmhsa_bench $shape $sw | grep us_per | sed 's/.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' || exit 1
  done
done
