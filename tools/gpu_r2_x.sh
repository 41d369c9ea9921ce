#!/bin/bash
# final tree of round 2: rocprofv3 kernel statistics of the headline bench command (no extra legs)
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
ARGS="bench.py --steps 5 --warmup 2 --no-asg --no-stress --no-c4 --no-c3 --no-c5 --no-cpu-baseline"
bash tools/prof.sh r2x_bench $ARGS; echo "prof rc=$?"
tail -2 gpurun_out/r2x_bench_run.log | cut -c1-600
head -24 gpurun_out/r2x_bench_kernel_stats.csv 2>/dev/null | cut -c1-160
