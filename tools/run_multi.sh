# grids of several rounds: the planner's choice against each family forced (plan-forcing build in tools/ab/env):
#   bash tools/run_multi.sh > log       columns: family, estimate, measured
for shape in "64 25 512 40 20" "32 25 512 80 40" "16 25 512 100 50" "8 25 512 300 200" "32 12 1024 60 30" "24 12 1024 60 30" "20 12 1024 60 30" "16 12 1024 100 50" "12 12 1024 100 50" "8 12 1024 200 100" "4 12 1024 300 200" "8 25 1024 100 50" "12 25 1024 60 30" "3 25 2048 60 30" "8 12 2048 60 30" "4 12 2048 60 30" "2 12 4096 40 20" "4 16 4096 20 10" "32 2 4096 40 20" "1 16 8192 20 10" "1 25 8192 20 10" "16 16 512 200 100" "64 12 256 200 100" "16 12 512 300 200" "32 12 512 200 100" "5 12 1024 300 200" "6 16 1024 200 100" "3 12 2048 100 50"; do
  echo "== $shape"
  for e in "X=0" "CMHSA_NOWP=1" "CMHSA_WAVES=1 CMHSA_LPT=0" "CMHSA_WAVES=1 CMHSA_LPT=1000000"; do
  env $e LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*estimate/  \1 est/; s/{.*us_per_launch": \([0-9.]*\).*/\1 us/' | paste - - || exit 1
  done
done
