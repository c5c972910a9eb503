#!/bin/bash
# conv epilogues with the saturation flag in a register (no counted cold path inside the store loops) and the 7x7 flow conv with
# channel groups as the fastest thread index: cur = HEAD, fc7 = only the 7x7 kernel, "" = both
set -e
out=gpurun_out/r04_epilogue.txt
mkdir -p gpurun_out; rm -f $out
for rep in 1 2; do
  for v in ${VARIANTS:-_cur _fc7 ""}; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=sequence REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out
  done
done
