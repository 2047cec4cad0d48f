# one-wave-workgroup plans against the 4-wave analytic plan on grids of several rounds:  bash tools/run_w1.sh > log
for shape in "64 25 512 40 20" "32 25 512 80 40" "16 25 512 100 50" "32 12 1024 60 30" "24 12 1024 60 30" "20 12 1024 60 30" "16 12 1024 100 50" "12 12 1024 100 50" "8 12 2048 60 30" "4 12 2048 60 30" "2 12 4096 40 20" "4 16 4096 20 10" "1 16 8192 20 10" "16 12 256 200 100" "64 12 256 200 100"; do
  echo "== $shape"
  CMHSA_NOWP=1 LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*estimate/  \1 est/; s/"steps.*us_per_launch/us/; s/"frac.*//' | paste - - || exit 1
  CMHSA_WAVES=1 LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*estimate/  \1 est/; s/"steps.*us_per_launch/us/; s/"frac.*//' | paste - - || exit 1
done
