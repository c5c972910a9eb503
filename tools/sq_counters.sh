#!/bin/bash
# SQ counter passes over the benchmark's forward (single stream), summarised per kernel into <out>/sq_counters.csv.
#   bash tools/sq_counters.sh gpurun_out/<tag>
out=${1:-gpurun_out/sq}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$i" -- python3 bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-config3 --no-f16-leg --no-f32-leg --no-per-frame-leg --no-h2d-leg > "$out/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 tools/summarize_sq.py "$out" --csv "$out/sq_counters.csv"
