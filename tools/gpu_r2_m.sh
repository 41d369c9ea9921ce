#!/bin/bash
# kernel statistics of the conv_glu (BASELINE config 4) step
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
bash tools/prof.sh r2m_c4 tools/c4_step.py; echo "prof rc=$?"
head -28 gpurun_out/r2m_c4_kernel_stats.csv | cut -c1-150
