#!/bin/bash
# within-run A/B of GEMM variants: interleaved rounds on the same GPU
for round in 1 2; do
for v in build_ab/*.so; do
  echo "== $v round $round"
  W2L_HIP_SO=$PWD/$v python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | grep -E "fc1 s1|fc2 s3|4096"
done
done
