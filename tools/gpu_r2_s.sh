#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/gemm_c5.py 2>&1 | grep "M=\|q/k/v" | tee gpurun_out/r2s_gemm_c5_after.log
timeout 600 python -m pytest tests/test_gpu_nn.py -k "gemm or linear" -x -q 2>&1 | tail -4 | tee gpurun_out/r2s_tests.log
timeout 300 python -m pytest tests/test_gpu_criterion.py -k "folded" -x -q 2>&1 | tail -4 | tee -a gpurun_out/r2s_tests.log
timeout 300 python tools/c5_step.py 3 f32 16 nodrop 2>&1 | grep "c5\]" | tee gpurun_out/r2s_c5.log
