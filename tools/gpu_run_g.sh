#!/bin/bash
# GPU session G: parity, conv probe (pipelined K loop), FCC ring variants with L2-resident tail loads, fix-up by quadrant, bench
mkdir -p gpurun_out
tag=${1:-r12}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -12 gpurun_out/${tag}_tests.log | cut -c1-200
timeout 300 python tools/gpu_probe.py conv 2>&1 | grep "conv" | tee gpurun_out/${tag}_conv.log
: > gpurun_out/${tag}_fcc.log
for v in "" "W2L_FCC_RING=1" "W2L_FCC_RING=2" "W2L_FCC_ASM=1" "" "W2L_FCC_RING=1"; do
  env $v timeout 300 python tools/gpu_probe.py fccstream 2>&1 | grep fccstream >> gpurun_out/${tag}_fcc.log
done
cat gpurun_out/${tag}_fcc.log
timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | tee gpurun_out/${tag}_gemm.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json | cut -c1-1700
timeout 900 bash tools/prof.sh ${tag}_bench bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asg --no-stress
head -24 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-160
