#!/bin/bash
# round 2 profiling call: rocprofv3 kernel statistics + the two PMC traffic passes over the SAME bench command
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
ARGS="bench.py --steps 5 --warmup 2 --no-asg --no-stress --no-c4 --no-c3 --no-cpu-baseline"
bash tools/prof.sh r2p_bench $ARGS; echo "prof rc=$?"
bash tools/pmc.sh r2p_fetch FETCH_SIZE $ARGS; echo "fetch rc=$?"
bash tools/pmc.sh r2p_write WRITE_SIZE $ARGS; echo "write rc=$?"
python tools/pmc_traffic.py gpurun_out/r2p_fetch_pmc.csv gpurun_out/r2p_write_pmc.csv gpurun_out/r2p_pmc_traffic.json | cut -c1-1200
head -30 gpurun_out/r2p_bench_kernel_stats.csv 2>/dev/null | cut -c1-200
# SQ counters of the role-swapped TDS convolution (instruction mix, MFMA busy)
bash tools/pmc.sh r2p_conv_sq "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" tools/conv_one.py; echo "conv sq rc=$?"
cat gpurun_out/r2p_conv_sq_pmc.csv 2>/dev/null | cut -c1-400
