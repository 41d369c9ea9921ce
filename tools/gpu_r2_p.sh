#!/bin/bash
mkdir -p gpurun_out
for sk in 1 0; do
  echo "== probe library, W2L_GEMM_SK=$sk"
  W2L_GEMM_SK=$sk timeout 200 python tools/gemm_c5.py --probe 2>&1 | grep "M=\|q/k/v"
done | tee gpurun_out/r2p_gemm_c5_sk.log
