#!/bin/bash
# forward milliseconds per pair against the clip length B (continued clips: what bench.py times), default library
set -e
out=gpurun_out/r04_clip_sweep.txt
mkdir -p gpurun_out; rm -f $out
for v in _base _m3 ""; do
  ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so B=16 MODE=continued REPS=10 python tools/stage_profile.py "lib$v" | tee -a $out
done
for b in ${BS:-12 14 15 16 17 18 20 24 32 34}; do
  B=$b MODE=continued REPS=6 python tools/stage_profile.py "B=$b" | tee -a $out
done
python - <<'PY' | tee -a gpurun_out/r04_clip_sweep.txt
import re
for line in open("gpurun_out/r04_clip_sweep.txt"):
    m = re.match(r"B=(\d+)\s+total ([0-9.]+) ms", line)
    if m:
        b, t = int(m.group(1)), float(m.group(2))
        print("B=%2d  %.3f ms per pair  (%.1f pairs/s one stream, no overlap)" % (b, t / b, 1e3 * b / t))
PY
