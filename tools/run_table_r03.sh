# every row of DESIGN.md section 3.4 from the shipped library, warmed up (the shader clock rises over the first ~20 ms):
#   bash tools/run_table_r03.sh      (on the GPU box; ~2 minutes)
run() {  # B H T steps warmup
  echo -n "B$1 H$2 T$3: "; timeout -k 5 100 tools/cmhsa_bench $1 $2 $3 $4 $5 2>&1 | sed 's/plan: \([^:]*\):.*/\1 |/; s/{.*us_per_launch": \([0-9.]*\).*tflops": \([0-9.]*\).*frac_of_157.3": \([0-9.]*\).*/\1 us \2 TF \3/' | paste - - || exit 1
}
for T in 8 64 128 256 384 512 640 768; do run 1 12 $T 4000 4000; done
for H in 12 16 20 25; do run 1 $H 1024 4000 4000; done
for T in 1536 2048 3072 4096 6144; do run 1 12 $T 500 2000; done
run 1 16 2048 500 2000; run 1 16 4096 300 1000; run 1 25 1536 500 2000; run 1 25 4096 300 1000; run 1 16 8192 100 300
for B in 2 4 5 8 12 16 24 32; do run $B 12 1024 500 2000; done
for B in 8 16 32 64; do run $B 25 512 500 2000; done
run 37 25 511 500 2000; run 64 25 256 500 2000
run 16 16 512 500 2000; run 4 16 2048 300 1000; run 32 2 4096 100 300; run 32 16 4096 20 20
