#!/bin/bash
# Feasibility probe for overlapping the lookup's sampling phase with convc1's MFMA phase:
# Build: git apply tools/ab/r04_lookup_probe.patch; python -m atdn_vslam_amd.build --variant lkp1 -DATDN_LOOKUP_PROBE=1; ... lkp2 -DATDN_LOOKUP_PROBE=2
# lkp1 = sampling loop with one K chunk (12 MFMAs, 8 LDS fragment reads, 2 weight loads) interleaved per unit and NO MFMA phase
# behind the barrier; lkp2 = the same plus three chunks left behind the barrier. Results are wrong by construction: timing only.
set -e
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "" _lkp1 _lkp2; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=sequence REPS=10 python tools/stage_profile.py "lib$v" | tee -a gpurun_out/r04_lookup_probe.txt
  done
done
