#!/bin/bash
# L2 hit / miss counters of the GEMM kernels in the bench step
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
ARGS="bench.py --steps 2 --warmup 1 --no-asg --no-stress --no-c4 --no-c3 --no-cpu-baseline"
bash tools/pmc.sh r2k_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $ARGS; echo "rc=$?"
head -8 gpurun_out/r2k_l2_pmc.csv | cut -c1-260
tail -5 gpurun_out/r2k_l2_pmc_run.log | cut -c1-200
