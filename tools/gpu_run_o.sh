#!/bin/bash
# GPU session O: parity (checkpoint resume), GEMM ablations on the buffer-addressed kernel
mkdir -p gpurun_out
tag=${1:-r23}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log | cut -c1-200
: > gpurun_out/${tag}_gemm_abl.log
for abl in 0 1 8 9 64 72 0; do
  W2L_GEMM_ABLBUF=$abl timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | sed "s/^/[ablbuf=$abl] /" >> gpurun_out/${tag}_gemm_abl.log
done
grep -v 8192 gpurun_out/${tag}_gemm_abl.log
