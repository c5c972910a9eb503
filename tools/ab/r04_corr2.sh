#!/bin/bash
# round 4: corr_bricks_kernel variants in ONE gpurun call — v2 (default: inline-asm tile loads, counted vmcnt), v0 (compiler
# loads: vmcnt(0) before every tile's LDS writes), v1 (v0 without result stores: timing only).
out=gpurun_out/r04_corr2
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -x -q -m gpu > $out/parity.txt 2>&1 || { tail -30 $out/parity.txt; exit 1; }
tail -2 $out/parity.txt
for rep in 1 2; do
  for v in v0 v1; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip_$v.so B=16 MODE=continued REPS=5 python3 tools/stage_profile.py $v >> $out/stages.txt 2>> $out/stages.err || exit 1
  done
  B=16 MODE=continued REPS=5 python3 tools/stage_profile.py v2 >> $out/stages.txt 2>> $out/stages.err || exit 1
done
cut -c1-75 $out/stages.txt
