#!/bin/bash
# SQ counter passes over tools/microbench_conv.py (diagnostic). Usage on the GPU box: bash tools/pmc_microbench.sh <outdir>
out=${1:-gpurun_out/pmc_mb}
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out/p$i" -- python3 tools/microbench_conv.py > "$out/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 tools/summarize_sq.py "$out"
