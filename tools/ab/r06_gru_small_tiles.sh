#!/bin/bash
# The ConvGRU gate convolutions at 16 pairs on 4 x 16-pixel x 64-channel blocks (120-126 registers: four waves per SIMD, no spill)
# against the shipped 8 x 16 x 128 blocks (two waves per SIMD); one job, alternating.
out=gpurun_out/r06/ab_gru_small_tiles2.txt
mkdir -p gpurun_out/r06
: > $out
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-f32-leg --no-per-frame-leg --no-h2d-leg"
for rep in 1 2; do
  for f in 0 2; do
    ATDN_CONV_SMALL_TILES=$f python3 bench.py --steps 12 --warmup 4 $LEGS > /tmp/ab.json 2>/dev/null
    python3 - $f $rep >> $out <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
s = d["stages_ms_per_forward"]
print("rep %s  gate blocks %s:  %.1f pairs/s   z|r %.3f + %.3f ms, q %.3f + %.3f ms per forward (one stream, eager)" % (
    sys.argv[2], {"0": "off (8x16x128)", "1": "4x16x64", "2": "4x16x128"}[sys.argv[1]], d["value"], s["gru_zr"], s["gru_zr_v"], s["gru_q"], s["gru_q_v"]))
PY
  done
done
cat $out
