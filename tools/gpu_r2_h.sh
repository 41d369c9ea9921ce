#!/bin/bash
# LDS / issue counters of the third-generation TDS convolution kernel (probe library), two passes per shape
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for c in 10 18; do
  bash tools/pmc.sh r2h_rs3_c${c}_a "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" tools/conv_one3.py $c
  bash tools/pmc.sh r2h_rs3_c${c}_b "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" tools/conv_one3.py $c
  grep -h "rs3\|^kernel" gpurun_out/r2h_rs3_c${c}_a_pmc.csv gpurun_out/r2h_rs3_c${c}_b_pmc.csv | cut -c1-400
done
