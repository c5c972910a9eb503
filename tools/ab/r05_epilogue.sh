#!/bin/bash
# round 5: stride-2 halo kernel + lean split-f16 store / load (cvt_pk, fma_mix) + weight scale folded into the epilogue FMAs +
# scalar-base addressing. Bit checks first, then the GPU parity suite, then stage times of base vs current in one job.
out=gpurun_out/r05_epilogue.txt
mkdir -p gpurun_out; rm -f $out
./tools/diag/sf_mix_check.bin 2>&1 | tee -a $out
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee -a $out
for rep in 1 2; do
  for v in ${VARIANTS:-_r5base ""}; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=continued REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out
  done
done
