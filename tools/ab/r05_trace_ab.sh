#!/bin/bash
# kernel traces of the same forward with two builds (A/B per kernel): bash tools/ab/r05_trace_ab.sh <tag> <variant suffixes...>
tag=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  name=${v:-cur}
  out=gpurun_out/${tag}_$name
  mkdir -p $out
  export ATDN_LIB_PATH=$PWD/atdn_vslam_amd/libatdn_hip$v.so
  B=16 MODE=continued REPS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/stage_profile.py $name > $out/log.txt 2>&1
  k=$(find $out/trace -name '*kernel_stats.csv' | head -1)
  python3 tools/summarize_rocprof.py "$k" gpurun_out/${tag}_${name}_kernel_stats.csv
  rm -rf $out/trace
done
