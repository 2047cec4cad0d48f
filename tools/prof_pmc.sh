#!/bin/bash
# rocprofv3 counter passes over tools/cmhsa_bench (one pass per run: rocprofv3 does not split counters).
# usage: tools/prof_pmc.sh TAG B H T [steps]      -> gpurun_out/pmc_TAG/passN/*.csv
set -u
TAG=$1; B=$2; H=$3; T=$4; STEPS=${5:-20}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD"
P3="FETCH_SIZE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- "$ROOT/tools/cmhsa_bench" $B $H $T $STEPS 3 > "$OUT/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/pass$i.log"; exit 1; }
done
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$ROOT/tools/cmhsa_bench" $B $H $T 200 20 > "$OUT/trace.log" 2>&1 || { echo "trace failed"; exit 1; }
grep us_per_launch "$OUT/trace.log"
echo done
