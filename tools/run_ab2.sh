for i in 1 2 3; do for v in base qdirect; do
  for shape in "1 12 1024 3000 3000" "1 16 1024 2000 2000" "4 12 1024 300 300" "1 12 512 3000 3000"; do
   echo -n "$v $shape: "; LD_LIBRARY_PATH=tools/ab/$v timeout -k 5 60 tools/cmhsa_bench $shape 2>&1 | grep us_per | sed 's/.*us_per_launch": \([0-9.]*\).*o00": \([0-9.-]*\).*/\1 us o00=\2/' || exit 1
  done
done; done
