#!/bin/bash
# GPU session R: kernel statistics of the current step (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out
tag=${1:-r26}
timeout 900 bash tools/prof.sh ${tag}_bench bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asg --no-stress
head -45 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-150
