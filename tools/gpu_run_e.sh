#!/bin/bash
# GPU session E: FCC stream variants (peeled loops, ring, nontemporal), GEMM buffer-addressed LDS-DMA A/B
mkdir -p gpurun_out
tag=${1:-r10}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -3 gpurun_out/${tag}_tests.log
: > gpurun_out/${tag}_fcc.log
for v in "" "W2L_FCC_RING=1" "W2L_FCC_ABL=4" "W2L_FCC_ABL=3" "W2L_FCC_ABL=7" "W2L_FCC_RING=1 W2L_FCC_ABL=4" "" "W2L_FCC_RING=1"; do
  env $v timeout 300 python tools/gpu_probe.py fccstream 2>&1 | grep fccstream >> gpurun_out/${tag}_fcc.log
done
cat gpurun_out/${tag}_fcc.log
: > gpurun_out/${tag}_gemm_buf.log
for r in 1 2; do
  timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | sed 's/^/[buf=0] /' >> gpurun_out/${tag}_gemm_buf.log
  W2L_GEMM_BUF=1 timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | sed 's/^/[buf=1] /' >> gpurun_out/${tag}_gemm_buf.log
done
cat gpurun_out/${tag}_gemm_buf.log
W2L_GEMM_BUF=1 timeout 300 python -m pytest tests/test_gpu_nn.py -m gpu -x -q -k gemm 2>&1 | tail -3
W2L_GEMM_BUF=1 timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | sed 's/^/[buf=1] /' | tee gpurun_out/${tag}_gemm_buf_all.log
timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | sed 's/^/[buf=0] /' | tee -a gpurun_out/${tag}_gemm_buf_all.log
