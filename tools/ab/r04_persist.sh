#!/bin/bash
# round 4, ONE gpurun call: persistent tiles for the thin 3x3 layers (libatdn_hip_persist.so: every 4-/6-wave 3x3 launch;
# _persist4: layers of <= 4 chunks only) against the default build; microbenchmark ladder first.
out=gpurun_out/r04_persist2
mkdir -p $out
export TMPDIR=/tmp
D=$PWD/atdn_vslam_amd
NIMG=16 python3 tools/microbench_conv_thin.py > $out/mb_thin16.txt 2>&1; grep -v "only:\|minus\|stores to" $out/mb_thin16.txt
ATDN_LIB_PATH=$D/libatdn_hip_persist.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -m gpu > $out/parity_persist.txt 2>&1 || { tail -30 $out/parity_persist.txt; exit 1; }
tail -2 $out/parity_persist.txt
for rep in 1 2; do
  B=16 MODE=continued REPS=5 python3 tools/stage_profile.py default >> $out/stages.txt 2>> $out/stages.err || exit 1
  for v in base persist persist4; do
    ATDN_LIB_PATH=$D/libatdn_hip_$v.so B=16 MODE=continued REPS=5 python3 tools/stage_profile.py $v >> $out/stages.txt 2>> $out/stages.err || exit 1
  done
done
python3 - <<'PY'
import re
for l in open("gpurun_out/r04_persist2/stages.txt"):
    m = dict(re.findall(r"(\w+) ([0-9.]+)", l.split("|")[1]))
    print("%-10s total %s fnet %s cnet %s motion %s mask %s flow_head %s" % (l.split()[0], l.split()[2], m["fnet"], m["cnet"], m["motion_encoder"], m["mask"], m["flow_head"]))
PY
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_default_$rep.json 2>> $out/bench.err || exit 1
  for v in base persist persist4; do
    ATDN_LIB_PATH=$D/libatdn_hip_$v.so python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_${v}_$rep.json 2>> $out/bench.err || exit 1
  done
done
grep -H -o '"value": [0-9.]*' $out/bench_*.json
