#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_attention.py -x -q 2>&1 | tail -3
bash tools/prof.sh r2w_c5 tools/c5_step.py 2 f32 16 nodrop; grep "c5\]" gpurun_out/r2w_c5_run.log
grep "bgemm\|softmax" gpurun_out/r2w_c5_kernel_stats.csv | cut -c1-120
