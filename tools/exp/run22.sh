python -m pytest tests/test_gpu_nn.py -m gpu -x -q -k "weight_and_bias or random_shapes" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 > gpurun_out/r06_run22_tests.log
python tools/gemm_dw_bias.py > gpurun_out/r06_run22_dw_bias.log 2>&1
python bench.py --no-asg --no-stress --no-c3 --no-c4 --no-c5 --no-cpu-baseline --no-train-binary --no-input-pipeline > gpurun_out/r06_run22_bench.json 2> gpurun_out/r06_run22_bench.err
