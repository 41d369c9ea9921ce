#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_fl_compat.py -k "lr_decay or data_parallel" -x -q 2>&1 | tail -25 | tee gpurun_out/r2z_tests.log
