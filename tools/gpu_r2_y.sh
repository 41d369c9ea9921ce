#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_trainer.py -k "torch_optim or adadelta_state" -x -q 2>&1 | tail -12 | tee gpurun_out/r2y_tests.log
