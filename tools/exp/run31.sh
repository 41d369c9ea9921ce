python -m pytest tests/test_gpu_asg_small.py tests/test_gpu_criterion.py tests/test_gpu_criterion_fuzz.py tests/test_gpu_parity_shapes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/r06_run31_tests.log
bash tools/prof.sh r06_run31_asg tools/asg_leg.py > gpurun_out/r06_run31_asg_run.log 2>&1
python tools/asg_leg.py > gpurun_out/r06_run31_asg_leg.json 2>/dev/null
