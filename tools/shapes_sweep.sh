cd tools
for s in "1 12 1024" "1 16 1024" "1 20 1024" "1 25 1024" "1 12 512" "1 12 256" "1 12 768" "2 12 1024" "4 12 1024" "16 12 1024" "1 1 8192" "32 2 4096"; do
  for mode in ws nows; do
    if [ "$mode" = "ws" ]; then extra=""; else extra="0 nows"; fi
    st=2000; wu=2000
    case "$s" in "16 12 1024"|"1 1 8192") st=200; wu=200;; "32 2 4096") st=40; wu=20;; esac
    out=$(timeout -k 5 100 ./cmhsa_bench $s $st $wu $extra 2>&1 | tr '\n' ' ')
    echo "$s $mode: $(echo $out | sed 's/.*plan: \([a-z ,]*\):.*us_per_launch": \([0-9.]*\), "tflops": \([0-9.]*\), "frac_of_157.3": \([0-9.]*\).*/\2 us \3 TF \4 | \1/')"
  done
done
