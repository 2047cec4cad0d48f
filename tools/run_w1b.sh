# borderline shapes: planner's own choice / one-wave workgroups forbidden / forced
for shape in "16 16 512 200 100" "32 8 512 200 100" "8 32 512 200 100" "4 16 2048 100 50" "8 16 1024 100 50" "16 8 1024 100 50" "2 16 4096 40 20" "8 25 512 300 200" "24 12 1024 60 30" "16 12 1024 100 50" "12 25 1024 60 30" "1 25 8192 20 10" "3 25 2048 60 30"; do
  echo "== $shape"
  for e in "X=0" "CMHSA_NOWP=1" "CMHSA_WAVES=1"; do
  env $e LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*estimate/  \1 est/; s/"steps.*us_per_launch/us/; s/"frac.*//' | paste - - || exit 1
  done
done
