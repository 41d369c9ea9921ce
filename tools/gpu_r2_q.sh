#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fl_compat.py -k "adagrad or adadelta or train_binary_reads or update" -x -q 2>&1 | tail -30 > gpurun_out/r2q_tests.log
cat gpurun_out/r2q_tests.log
