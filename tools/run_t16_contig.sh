for shape in "1 12 128" "1 12 256" "1 12 384" "1 12 512" "1 12 640" "1 16 512" "1 20 448" "1 20 512" "1 25 256" "1 25 384" "1 25 512" "8 12 128" "1 12 768" "1 12 1024"; do
  for mode in "CMHSA_T16_CONTIG=0" "CMHSA_T16_CONTIG=1"; do
    echo -n "$shape $mode: "; env CMHSA_T16=1 CMHSA_T16_SUB=1 CMHSA_T16_WAVES=4 $mode timeout -k 5 60 tools/cmhsa_bench_env $shape 1500 2000 2>&1 | grep us_per | sed 's/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' || exit 1
  done
done
