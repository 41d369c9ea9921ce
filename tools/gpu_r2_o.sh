#!/bin/bash
# config 5: timing (dropout on / off, f32 / bf16) and kernel statistics of the dropout-free f32 step
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
for args in "3 f32 16 nodrop" "3 f32 16 drop" "3 bf16 16 nodrop"; do
  timeout 300 python tools/c5_step.py $args 2>&1 | grep "c5\]\|Error\|error" | tee -a gpurun_out/r2o_c5.log
done
bash tools/prof.sh r2o_c5 tools/c5_step.py 2 f32 16 nodrop; echo "prof rc=$?"
head -32 gpurun_out/r2o_c5_kernel_stats.csv | cut -c1-160
