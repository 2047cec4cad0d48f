#!/bin/bash
# Regenerates EVERY measurement file under profiles/ for one round from ONE build, in one GPU session:
#   tools/prof_all.sh r02        (run on the GPU box, e.g. gpurun -- 'bash tools/prof_all.sh r02')
# writes gpurun_out/prof_<round>/...; tools/collect_profiles.py <round> then copies the summaries into profiles/.
# Every GPU step has its own time limit; the script stops at the first failure.
set -u
ROUND=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$ROUND
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
step() { echo "== $*" | tee -a "$OUT/steps.log"; }
fail() { echo "FAILED: $*" | tee -a "$OUT/steps.log"; exit 1; }

step "bench.py default (config 2 headline, other workloads, generation, cpu baseline)"
timeout -k 10 400 python3 bench.py > "$OUT/bench_config2.json" 2> "$OUT/bench_config2.err" || fail bench_config2
step "bench.py config5_shard / config5 on one GPU"
timeout -k 10 200 python3 bench.py --workload config5_shard --steps 50 --warmup 10 --no-extras > "$OUT/bench_config5_shard.json" 2> "$OUT/bench_config5_shard.err" || fail bench_config5_shard
timeout -k 10 200 python3 bench.py --workload config5 --steps 20 --warmup 5 --no-extras > "$OUT/bench_config5_n1.json" 2> "$OUT/bench_config5_n1.err" || fail bench_config5_n1

cd /tmp && export TMPDIR=/tmp
step "rocprofv3 kernel trace of the bench.py command (headline only: every launch is config 2)"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench_py" -- python3 "$ROOT/bench.py" --no-extras --no-cpu-baseline > "$OUT/trace_bench_py.json" 2> "$OUT/trace_bench_py.err" || fail trace_bench_py
step "rocprofv3 kernel trace of tools/cmhsa_bench (config 2 from a cold device, config-5 shard)"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_config2" -- "$ROOT/tools/cmhsa_bench" 1 12 1024 4000 4000 > "$OUT/trace_config2.log" 2>&1 || fail trace_config2
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_config5_shard" -- "$ROOT/tools/cmhsa_bench" 32 2 4096 50 10 > "$OUT/trace_config5_shard.log" 2>&1 || fail trace_config5_shard
step "rocprofv3 kernel trace: the B = 1 launches of a generation (config 3: 16-key-tile kernel) and config 4's attention shape (job lists)"
for CFG in "config3_t128 1 12 128 4000 4000" "config3_t256 1 12 256 4000 4000" "config3_t512 1 12 512 4000 4000" "config4_shape 64 25 512 300 2000"; do
  set -- $CFG
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$1" -- "$ROOT/tools/cmhsa_bench" $2 $3 $4 $5 $6 > "$OUT/trace_$1.log" 2>&1 || fail trace_$1
done

# counter passes: one --pmc pass per run (rocprofv3 does not split counters), --pmc never combined with a trace
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD"
P3="FETCH_SIZE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
for CFG in "config2 1 12 1024 50" "config5_shard 32 2 4096 10" "config3_t256 1 12 256 50" "config4_shape 64 25 512 10"; do
  set -- $CFG
  TAG=$1
  step "rocprofv3 --pmc passes, $TAG"
  mkdir -p "$OUT/pmc_$TAG"
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc_$TAG/pass$i" -- "$ROOT/tools/cmhsa_bench" $2 $3 $4 $5 3 > "$OUT/pmc_$TAG/pass$i.log" 2>&1 || fail "pmc $TAG pass $i"
  done
done
step "all done"
