#!/bin/bash
# (1) effective clock per kernel (GRBM_GUI_ACTIVE / 8 / wall time), one stream; (2) A/B: N = 192 layers on the 96-wide block
set -e
export TMPDIR=/tmp
out=gpurun_out/r04_clock
mkdir -p $out
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/raw -- python3 bench.py --steps 2 --warmup 1 --streams 1 $LEGS > $out/bench.log 2>&1
f=$(find $out/raw -name '*counter_collection.csv' | head -1)
python3 tools/summarize_clock.py "$f" $out/clock.csv | tee $out/clock.txt
rm -rf $out/raw
for rep in 1 2; do
  for v in "" _bn96; do
    ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=sequence REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out/bn96.txt
  done
done
