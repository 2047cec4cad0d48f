# round-3 baseline: in-kernel stamps of the small grids of config 3 (B=1 H=12, T = 128/256/512/768) and their
# steady-state times:  bash tools/run_r03_baseline.sh   (writes gpurun_out/r03_baseline/)
set -u
OUT=gpurun_out/r03_baseline; mkdir -p $OUT
timeout -k 5 60 tools/mfma16_probe > $OUT/mfma16_probe.txt 2>&1 || { cat $OUT/mfma16_probe.txt; exit 1; }
cat $OUT/mfma16_probe.txt
for T in 64 128 256 384 512 768 1024; do
  echo -n "B1 H12 T=$T: "; timeout -k 5 60 tools/cmhsa_bench 1 12 $T 2000 3000 2>&1 | sed 's/plan: \([^:]*\):.*\(SIMD cap [0-9]*\).*/\1, \2 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*/\1 us \2 TF/' | paste - - || exit 1
done | tee $OUT/times.txt
for T in 128 256 512 768; do
  timeout -k 5 60 tools/cmhsa_bench_stamps 1 12 $T 20 3000 0 > $OUT/stamps_b1_h12_t$T.txt 2>&1 || exit 1
done
echo done
