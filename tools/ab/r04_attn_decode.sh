#!/bin/bash
# attention x V: the H3 decode rewritten for instruction count (d0: compiler's choice of FMA, d1: plain v_fma_f32 forced,
# d2: v_fma_mixlo/mixhi_f16) and pinned to its memory phase (d0np: not pinned), against the build of HEAD (base);
# m0/m1/m2: the decode of chunk q + 1 inside the multiply of chunk q (same wave, in the MFMA gaps), FMA forms as above
set -e
out=gpurun_out/r04_attn_decode.txt
mkdir -p gpurun_out
for rep in 1 2; do
  for v in ${VARIANTS:-_base _d1 _m0 _m1 _m2}; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=sequence REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out
  done
done
for v in ${TESTV:-_m0}; do
  ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "attn or attention or h3 or H3 or aggregate" 2>&1 | tail -3 | tee -a $out
done
