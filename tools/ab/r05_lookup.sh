#!/bin/bash
# lookup sampling unit, round 5: base = the round-4 kernel (18 chains through an LDS table), "" = the working tree's build.
# Parity gate first (tests that see the lookup), then stage times of both builds in one job.
out=gpurun_out/r05_lookup.txt
mkdir -p gpurun_out; rm -f $out
python -m pytest tests/test_gpu_parity.py -q -k "large_batch" 2>&1 | tail -3 | tee -a $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -q 2>&1 | tail -8 | tee -a $out
for rep in 1 2; do
  for v in ${VARIANTS:-_r5base ""}; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=continued REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out
  done
done
