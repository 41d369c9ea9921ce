#!/bin/bash
# GPU session H: parity (256x128 three-stage GEMM), GEMM A/B, conv ablations, bench
mkdir -p gpurun_out
tag=${1:-r13}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -12 gpurun_out/${tag}_tests.log | cut -c1-200
for r in 1 2; do
  timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | sed 's/^/[p3=1] /' | tee -a gpurun_out/${tag}_gemm.log
  W2L_GEMM_P3=0 timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | sed 's/^/[p3=0] /' | tee -a gpurun_out/${tag}_gemm.log
done
W2L_GEMM_P3=2 timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | sed 's/^/[p3=2] /' | tee -a gpurun_out/${tag}_gemm.log
for a in 0 1 2 4 6 7; do
  W2L_TDS_ABL=$a timeout 300 python tools/gpu_probe.py conv 2>&1 | grep "conv\] tds" | sed "s/^/[abl=$a] /" | tee -a gpurun_out/${tag}_conv_abl.log
done
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json | cut -c1-1300
W2L_GEMM_P3=0 timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-asg --no-stress > gpurun_out/${tag}_bench_p3off.json 2>> gpurun_out/${tag}_bench.err
cat gpurun_out/${tag}_bench_p3off.json | cut -c1-700
