# HBM traffic and L2 hit rate of B64 H25 T512, job lists against wave pairs:   bash tools/pmc_list.sh   (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_list
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in 0 1; do
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    CMHSA_LIST=$mode timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d "$OUT/list${mode}_pass$i" -- "$ROOT/tools/cmhsa_bench_env" 64 25 512 10 3 > "$OUT/list${mode}_pass$i.log" 2>&1 || exit 1
  done
done
