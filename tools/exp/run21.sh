python -m pytest tests/test_gpu_nn.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -15 > gpurun_out/r06_run21_tests.log
python bench.py --no-asg --no-stress --no-c3 --no-c4 --no-c5 --no-cpu-baseline --no-train-binary --no-input-pipeline > gpurun_out/r06_run21_bench.json 2> gpurun_out/r06_run21_bench.err
