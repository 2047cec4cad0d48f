# one wave per block, longest first within groups of G (b,h)s, against the planner's choice:  bash tools/run_lpt.sh
for shape in "64 25 512 40 20" "32 16 4096 6 3" "32 2 4096 40 20" "16 12 1024 100 50" "32 12 1024 60 30" "8 12 2048 60 30"; do
  echo "== $shape"
  env X=0 LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*estimate/  \1 est/; s/"steps.*us_per_launch/us/; s/"frac.*//' | paste - - || exit 1
  for G in 8 16 32 64 128 256 100000; do
  echo -n "  G=$G "; env CMHSA_WAVES=1 CMHSA_LPT=$G LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | grep us_per | sed 's/.*us_per_launch/us/; s/"frac.*//' || exit 1
  done
done
