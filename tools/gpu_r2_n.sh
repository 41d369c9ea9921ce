#!/bin/bash
# transformer bring-up: op tests + end-to-end
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_trainer.py -k "attention or transformer or pool or batched or relative" -x -q 2>&1 | tail -40 > gpurun_out/r2n_tests.log
cat gpurun_out/r2n_tests.log
