#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_trainer.py -k "attention or transformer or pool or batched or relative or padding" -x -q 2>&1 | tail -30 > gpurun_out/r2v_tests.log
cat gpurun_out/r2v_tests.log
