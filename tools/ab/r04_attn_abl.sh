#!/bin/bash
# attention x V ablations (diagnostic builds, wrong results, timing only): ATDN_ATTN_ABL bit 0 = no MFMAs, bit 1 = no H3 decode,
# bit 2 = default cache policy instead of non-temporal loads, bit 3 = V^T fragments read from LDS once, bit 4 = no V^T staging in
# the loop, bit 5 = no block barriers in the loop
set -e
out=gpurun_out/r04_attn_abl.txt
mkdir -p gpurun_out; rm -f $out
for rep in 1 2; do
  for v in "" ${VARIANTS:-_aabl3 _aabl11 _aabl19 _aabl35 _aabl27 _aabl59 _aabl32}; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=sequence REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out
  done
done
