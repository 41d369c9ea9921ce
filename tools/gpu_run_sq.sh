#!/bin/bash
# MFMA utilisation counters of the current GEMM kernels (separate --pmc passes, kernel-trace only)
mkdir -p gpurun_out
tag=${1:-r53}
for shape in "24000 2400 800 fwd" "6016 1440 4320 dx" "4096 4096 4096 fwd"; do
  n=$(echo $shape | tr ' ' '_')
  timeout 300 bash tools/pmc.sh ${tag}_sq1_$n "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" tools/gemm_one.py $shape
  timeout 300 bash tools/pmc.sh ${tag}_sq2_$n "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" tools/gemm_one.py $shape
done
grep -h "gemm1" gpurun_out/${tag}_sq*_pmc.csv | cut -c1-300
grep -h "kernel," gpurun_out/${tag}_sq1_*pmc.csv | head -1; grep -h "kernel," gpurun_out/${tag}_sq2_*pmc.csv | head -1
