#!/bin/bash
# kernel statistics of the streaming TDS (BASELINE config 3) step, fp32 + bf16 legs
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
ARGS="bench.py --steps 1 --warmup 1 --no-asg --no-stress --no-c4 --no-cpu-baseline"
bash tools/prof.sh r2l_c3 $ARGS; echo "prof rc=$?"
head -40 gpurun_out/r2l_c3_kernel_stats.csv | cut -c1-150
