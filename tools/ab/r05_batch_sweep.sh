#!/bin/bash
# Clip length (frame pairs per step) sweep of bench.py's headline loop on one box: value per --batch, two passes in alternating order.
# usage (GPU box): bash tools/ab/r05_batch_sweep.sh OUT  ->  gpurun_out/OUT.txt
set -e
out=gpurun_out/${1:-r05_batch_sweep}.txt
mkdir -p gpurun_out; : > $out
for pass in 1 2; do
  for B in ${BATCHES:-16 17 18 20 24 32}; do
    timeout -k 10 300 python bench.py --batch $B --steps $((320 / B)) --warmup 4 --no-cpu-baseline --no-h2d-leg --no-config3 --no-f16-leg \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch %3d  %.1f pairs/s  %.2f ms/step' % ($B, d['value'], d['ms_per_step']))" | tee -a $out
  done
done
