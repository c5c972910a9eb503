#!/bin/bash
# the parallel graph branches of the low-latency form (feature chain beside context chain, flow branch beside
# correlation branch) forced on at 16 pairs per launch, against the single chain; bench.py's two streams, one job, alternating.
out=gpurun_out/r06/ab_par_branches.txt
mkdir -p gpurun_out/r06
: > $out
LEGS="--no-cpu-baseline --no-f16-leg --no-f32-leg --no-per-frame-leg --no-h2d-leg"
for rep in 1 2; do
  for f in 0 1; do
    ATDN_LOW_LATENCY=1 ATDN_ATTN_FORCE_SPLIT=1 ATDN_PAR_FORCE=$f python3 bench.py --steps 12 --warmup 4 $LEGS > /tmp/ab.json 2>/dev/null
    python3 - $f $rep >> $out <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print("rep %s  branches %s:  %.1f pairs/s (two streams)   config3 (two lanes) %.1f pairs/s" % (sys.argv[2], "on " if sys.argv[1] == "1" else "off", d["value"], d["config3"]["value"]))
PY
  done
done
cat $out
