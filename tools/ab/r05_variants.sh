#!/bin/bash
# stage times of several builds in one job: VARIANTS="_r5base _oldall ''" bash tools/ab/r05_variants.sh <out name>
out=gpurun_out/${1:-r05_variants}.txt
mkdir -p gpurun_out; rm -f $out
for rep in 1 2; do
  for v in ${VARIANTS}; do
    [ "$v" = "cur" ] && v=""
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=continued REPS=8 python tools/stage_profile.py "lib$v" | tee -a $out
  done
done
