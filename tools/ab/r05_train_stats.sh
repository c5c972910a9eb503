#!/bin/bash
# Trainer: BatchNorm statistics fused into the producing kernels against the separate reduction passes (ATDN_TRAIN_FUSED_STATS=0),
# same library, same job, alternating. usage (GPU box): bash tools/ab/r05_train_stats.sh OUT -> gpurun_out/OUT.txt
out=gpurun_out/${1:-r05_train_stats}.txt
mkdir -p gpurun_out; : > $out
for pass in 1 2; do
  for v in 1 0; do
    ATDN_TRAIN_FUSED_STATS=$v timeout -k 10 300 python tools/bench_train.py 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_stats=$v  %.3f ms per iteration  %.1f flow frames/s  loss %.6f -> %.6f' % (d['ms_per_iteration'], d['value'], d['loss_first'], d['loss_last']))" | tee -a $out
  done
done
