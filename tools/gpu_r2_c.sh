#!/bin/bash
# round 2, call C: small-shape check of the role-swapped conv (the failing full-network gradient), ablation timings,
# the failing trainer test alone, Train binary test, bench with the fixed host legs
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 200 python tools/conv_rs.py --small > gpurun_out/r2c_conv_small.log 2>&1; echo "small rc=$?"; grep "^small" gpurun_out/r2c_conv_small.log; tail -3 gpurun_out/r2c_conv_small.log
timeout 300 python tools/conv_rs.py --abl > gpurun_out/r2c_conv_abl.log 2>&1; echo "abl rc=$?"; grep -v amdgpu gpurun_out/r2c_conv_abl.log | tail -45
timeout 600 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fl_compat.py -m gpu -q -k "config2 or train_binary" > gpurun_out/r2c_tests.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r2c_tests.log | cut -c1-300
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err; echo "bench rc=$?"
grep "^\[bench" gpurun_out/r2c_bench.err; tail -c 1500 gpurun_out/r2c_bench.json
