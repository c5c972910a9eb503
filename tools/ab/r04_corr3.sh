#!/bin/bash
# round 4: brick-major pyramid layout + corr_bricks_kernel (default build) vs the round-3 state (generic GEMM kernel for the
# correlation levels, a pixel's whole map contiguous: libatdn_hip_base.so built from the previous commit), one gpurun call.
out=gpurun_out/r04_corr5
mkdir -p $out
export TMPDIR=/tmp
BASE=$PWD/atdn_vslam_amd/libatdn_hip_base.so
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1 || { tail -30 $out/gpu_tests.txt; exit 1; }
tail -2 $out/gpu_tests.txt
for rep in 1 2; do
  ATDN_LIB_PATH=$BASE B=16 MODE=continued REPS=5 python3 tools/stage_profile.py base >> $out/stages.txt 2>> $out/stages.err || exit 1
  B=16 MODE=continued REPS=5 python3 tools/stage_profile.py new >> $out/stages.txt 2>> $out/stages.err || exit 1
done
cut -c1-110 $out/stages.txt
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
for rep in 1 2; do
  ATDN_LIB_PATH=$BASE python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_base_$rep.json 2>> $out/bench.err || exit 1
  python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_new_$rep.json 2>> $out/bench.err || exit 1
done
grep -H -o '"value": [0-9.]*' $out/bench_*.json
