#!/bin/bash
# round 4, A/B of two builds in ONE gpurun call: corr_bricks_kernel (default build) vs the generic GEMM kernel for the
# correlation levels (-DATDN_CORR_GENERIC, libatdn_hip_generic.so).
out=gpurun_out/r04_corr
mkdir -p $out
export TMPDIR=/tmp
G=$PWD/atdn_vslam_amd/libatdn_hip_generic.so
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu > $out/parity.txt 2>&1 || { tail -30 $out/parity.txt; exit 1; }
tail -3 $out/parity.txt
for rep in 1 2; do
  ATDN_LIB_PATH=$G B=16 MODE=continued REPS=5 python3 tools/stage_profile.py generic >> $out/stages.txt 2>> $out/stages.err || exit 1
  B=16 MODE=continued REPS=5 python3 tools/stage_profile.py corr_bricks >> $out/stages.txt 2>> $out/stages.err || exit 1
done
cut -c1-75 $out/stages.txt
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
for rep in 1 2; do
  ATDN_LIB_PATH=$G python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_generic_$rep.json 2>> $out/bench.err || exit 1
  python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_new_$rep.json 2>> $out/bench.err || exit 1
done
grep -H -o '"value": [0-9.]*' $out/bench_*.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 4 --warmup 2 --streams 1 $LEGS > $out/rocprof.log 2>&1
k=$(find $out/trace -name '*kernel_stats.csv' | head -1); python3 tools/summarize_rocprof.py "$k" $out/kernel_stats.csv > /dev/null; grep -i "corr_bricks\|brick_rows\|pool_features" $out/kernel_stats.csv | cut -c1-200
