#!/bin/bash
# attention x V cut along its key axis at 16 pairs per launch (the real kernel behind "stream-K" arithmetic):
# one-stream stage time of the aggregate and bench.py's two-stream rate, n = 1 (off) / 2 / 4 / 8 key ranges, alternating.
out=gpurun_out/r06/ab_attn_split.txt
mkdir -p gpurun_out/r06
: > $out
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-f32-leg --no-per-frame-leg --no-h2d-leg"
for rep in 1 2; do
  for n in 1 8 4 2; do
    ATDN_LOW_LATENCY=1 ATDN_ATTN_FORCE_SPLIT=$n python3 bench.py --steps 12 --warmup 4 $LEGS > /tmp/ab.json 2>/dev/null
    python3 - $n $rep >> $out <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print("rep %s  key ranges %s:  %.1f pairs/s (two streams)   aggregate stage %.3f ms per forward (one stream, eager)" % (sys.argv[2], sys.argv[1], d["value"], d["stages_ms_per_forward"]["aggregate"]))
PY
  done
done
cat $out
