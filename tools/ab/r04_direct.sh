#!/bin/bash
# round 4, A/B in ONE gpurun call: channel-vector epilogues of the 16x16x32 halo kernels storing straight from the accumulator
# layout (-DATDN_SF6_DIRECT, libatdn_hip_direct.so) vs through the wave-private LDS transpose (default build).
out=gpurun_out/r04_direct
mkdir -p $out
export TMPDIR=/tmp
D=$PWD/atdn_vslam_amd/libatdn_hip_direct.so
ATDN_LIB_PATH=$D timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/parity_direct.txt 2>&1 || { tail -30 $out/parity_direct.txt; exit 1; }
tail -3 $out/parity_direct.txt
for rep in 1 2; do
  B=16 MODE=continued REPS=5 python3 tools/stage_profile.py default >> $out/stages.txt 2>> $out/stages.err || exit 1
  ATDN_LIB_PATH=$D B=16 MODE=continued REPS=5 python3 tools/stage_profile.py direct >> $out/stages.txt 2>> $out/stages.err || exit 1
done
cat $out/stages.txt
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_default_$rep.json 2>> $out/bench.err || exit 1
  ATDN_LIB_PATH=$D python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_direct_$rep.json 2>> $out/bench.err || exit 1
done
grep -H -o '"value": [0-9.]*' $out/bench_*.json
