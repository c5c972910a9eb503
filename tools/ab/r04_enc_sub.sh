#!/bin/bash
# round 4, A/B in ONE gpurun call: encoders depth-first in sub-batches of frames (ATDN_ENC_SUB) vs breadth-first.
# Per-stage HIP-event times at the bench's clip shape (16 pairs, continued clip), then bench.py on both settings.
out=gpurun_out/r04_enc_sub
mkdir -p $out
export TMPDIR=/tmp
python3 tools/ab/r04_enc_sub_bits.py > $out/bits.txt 2>&1 || { cat $out/bits.txt; exit 1; }
cat $out/bits.txt
for sub in 0 8 6 4 3 2 0 3; do
  ATDN_ENC_SUB=$sub B=16 MODE=continued REPS=5 python3 tools/stage_profile.py sub$sub >> $out/stages.txt 2>> $out/stages.err || exit 1
done
cat $out/stages.txt
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-h2d-leg"
for sub in 0 3 4 0 3 4; do
  ATDN_ENC_SUB=$sub python3 bench.py --steps 20 --warmup 3 $LEGS > $out/bench_sub${sub}_$RANDOM.json 2>> $out/bench.err || exit 1
done
grep -h -o '"value": [0-9.]*' $out/bench_sub*.json
ls $out
