#!/bin/bash
# round 4, ONE gpurun call: attention x V experiments (VERDICT r3 #5 a, b) as library variants, the corr kernel with compiler
# loads (corrv0), and the thin-conv ablation ladder with the "stores stay in L2" build.
out=gpurun_out/r04_attn
mkdir -p $out
export TMPDIR=/tmp
D=$PWD/atdn_vslam_amd
for v in split both; do
  ATDN_LIB_PATH=$D/libatdn_hip_$v.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flow or attention or gma" > $out/parity_$v.txt 2>&1 || { tail -20 $out/parity_$v.txt; exit 1; }
  tail -1 $out/parity_$v.txt
done
for rep in 1 2 3; do
  B=16 MODE=continued REPS=5 python3 tools/stage_profile.py default >> $out/stages.txt 2>> $out/stages.err || exit 1
  for v in prio split both corrv0; do
    ATDN_LIB_PATH=$D/libatdn_hip_$v.so B=16 MODE=continued REPS=5 python3 tools/stage_profile.py $v >> $out/stages.txt 2>> $out/stages.err || exit 1
  done
done
python3 - <<'PY'
import re
for l in open("gpurun_out/r04_attn/stages.txt"):
    m = dict(re.findall(r"(\w+) ([0-9.]+)", l.split("|")[1]))
    print("%-10s total %s aggregate %s corr %s pool %s lookup %s" % (l.split()[0], l.split()[2], m["aggregate"], m["corr"], m["pool"], m["lookup"]))
PY
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
for rep in 1 2; do
  python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_default_$rep.json 2>> $out/bench.err || exit 1
  for v in prio split both; do
    ATDN_LIB_PATH=$D/libatdn_hip_$v.so python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_${v}_$rep.json 2>> $out/bench.err || exit 1
  done
done
grep -H -o '"value": [0-9.]*' $out/bench_*.json
NIMG=16 python3 tools/microbench_conv_thin.py > $out/mb_thin16.txt 2>&1; head -14 $out/mb_thin16.txt
