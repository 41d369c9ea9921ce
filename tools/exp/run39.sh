python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fl_compat.py tests/test_gpu_attention.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/r06_run39_tests.log
echo "c3: $(python tools/c3_step.py 5 bf16 2>&1 | grep '\[c3\]' | cut -c1-230)" > gpurun_out/r06_run39_steps.log
echo "c5: $(python tools/c5_step.py 5 bf16 2>&1 | grep '\[c5\]' | cut -c1-230)" >> gpurun_out/r06_run39_steps.log
