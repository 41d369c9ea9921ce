#!/bin/bash
# round 2, call A: new TDS conv kernel probe, the full GPU test-suite, the default bench line
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 300 python tools/conv_rs.py > gpurun_out/r2a_conv_rs.log 2>&1
echo "conv_rs rc=$?" 
tail -25 gpurun_out/r2a_conv_rs.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a_tests.log 2>&1
echo "pytest rc=$?"
tail -15 gpurun_out/r2a_tests.log
timeout 600 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r2a_bench.json
