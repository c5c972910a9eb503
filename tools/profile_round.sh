#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: run on the GPU box from the repo root.
#   bash tools/profile_round.sh <tag>            (writes under gpurun_out/<tag>/)
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace2" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config3 --no-f16-leg --no-f32-leg --no-per-frame-leg --no-h2d-leg > "$out/bench_under_rocprof.json" 2> "$out/trace2.err"
# per-kernel passes: one stream, so that every kernel runs alone and its duration / counters are its own; secondary legs off
# (they time other things)
LEGS="--no-cpu-baseline --no-config3 --no-f16-leg --no-f32-leg --no-per-frame-leg --no-h2d-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace1" -- python3 bench.py --steps 10 --warmup 2 --streams 1 $LEGS > "$out/bench_under_rocprof_1stream.json" 2> "$out/trace1.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 bench.py --steps 2 --warmup 1 --streams 1 $LEGS > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 bench.py --steps 2 --warmup 1 --streams 1 $LEGS > "$out/pmc_write.log" 2>&1
f=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
w=$(find "$out/write" -name '*counter_collection.csv' | head -1)
python3 tools/summarize_pmc.py "$f" "$w" "$out/pmc"
for t in trace2 trace1; do
  k=$(find "$out/$t" -name '*kernel_stats.csv' | head -1)
  python3 tools/summarize_rocprof.py "$k" "$out/${t}_kernel_stats.csv"
done
ls -la "$out"
