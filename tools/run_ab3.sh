# A/B of two builds (tools/ab/<name>/libcmhsa_hip.so), interleaved three times: args = the two names
A=${1:-env}; B=${2:-qglobal}
for i in 1 2 3; do for v in $A $B; do
  for shape in "1 12 1024 3000 3000" "1 16 1024 2000 2000" "1 25 1024 2000 2000" "1 12 512 3000 3000" "1 12 256 3000 3000" "2 12 1024 1000 1000" "1 12 2048 500 500"; do
   echo -n "$v $shape: "; LD_LIBRARY_PATH=tools/ab/$v timeout -k 5 60 tools/cmhsa_bench $shape 2>&1 | grep us_per | sed 's/.*us_per_launch": \([0-9.]*\).*o00": \([0-9.-]*\).*/\1 us o00=\2/' || exit 1
  done
done; done
