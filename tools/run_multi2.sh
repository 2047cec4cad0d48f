# grids of a third of a round to one round of one-wave jobs: planner's choice / one-wave plans forbidden / pairs forced
for shape in "4 12 1024 300 200" "5 12 1024 300 200" "6 12 1024 300 200" "8 12 1024 200 100" "10 12 1024 200 100" "8 12 512 400 300" "10 12 512 400 300" "16 12 512 300 200" "2 12 2048 200 100" "2 16 2048 200 100" "1 25 2048 200 100" "1 20 4096 60 30" "4 25 512 400 300" "5 25 512 400 300" "2 25 1024 300 200" "3 20 1024 300 200" "1 12 4096 100 50" "1 16 4096 100 50" "7 12 768 300 200" "3 16 1536 200 100"; do
  echo "== $shape"
  for e in "X=0" "CMHSA_NOWP=1" "CMHSA_WAVES=1 CMHSA_LPT=0" "CMHSA_WAVES=1 CMHSA_LPT=1000000"; do
  env $e LD_LIBRARY_PATH=tools/ab/env timeout -k 5 120 tools/cmhsa_bench $shape 2>&1 | sed 's/plan: \([^:]*\):.*estimate/  \1 est/; s/{.*us_per_launch": \([0-9.]*\).*/\1 us/' | paste - - || exit 1
  done
done
