#!/bin/bash
# round 2, call B: RMW overlap-add conv kernel, the new fl:: C++ tests, the failing golden test, bench with progress notes
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 300 python tools/conv_rs.py > gpurun_out/r2b_conv_rs.log 2>&1
echo "conv_rs rc=$?"
tail -22 gpurun_out/r2b_conv_rs.log
timeout 600 python -m pytest tests/test_gpu_fl_compat.py tests/test_gpu_parity_shapes.py tests/test_gpu_trainer.py -m gpu -q > gpurun_out/r2b_tests.log 2>&1
echo "pytest rc=$?"
tail -30 gpurun_out/r2b_tests.log
timeout 420 python bench.py > gpurun_out/r2b_bench.json 2> gpurun_out/r2b_bench.err
echo "bench rc=$?"
grep "^\[bench" gpurun_out/r2b_bench.err
tail -c 2500 gpurun_out/r2b_bench.json
