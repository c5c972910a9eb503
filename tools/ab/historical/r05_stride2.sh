#!/bin/bash
# stride-2 3x3 encoder convolutions on the halo-patch kernel (S = 2): parity first, then stage times against the base build
out=gpurun_out/r05_stride2.txt
mkdir -p gpurun_out; rm -f $out
python -m pytest tests/test_gpu_parity.py -q -k "stride2 or c1_stages or c2_kitti or full_flow" 2>&1 | tail -4 | tee -a $out
python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "encoders or clip_modes or two_lane" 2>&1 | tail -4 | tee -a $out
for rep in 1 2; do
  for v in ${VARIANTS:-_r5base ""}; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=continued REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out
  done
done
