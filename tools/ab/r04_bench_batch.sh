#!/bin/bash
# bench.py headline (two streams) against the clip length, same job; plus the HEAD build at 16 for reference
set -e
out=gpurun_out/r04_bench_batch.txt
mkdir -p gpurun_out; rm -f $out
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
run() { python bench.py $LEGS --batch $1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2 batch $1: %.1f pairs/s, %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $out; }
if [ -f atdn_vslam_amd/libatdn_hip_base.so ]; then ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip_base.so run 16 base; fi   # (a build of an earlier commit, if present)
for rep in 1 2; do for b in ${BS:-16 17 34}; do run $b new; done; done
