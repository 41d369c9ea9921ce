python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/r06_run40_gpu_tests.log
bash tools/prof.sh r06_run40_c3 tools/c3_step.py 3 bf16 > /dev/null 2>&1
bash tools/prof.sh r06_run40_c5 tools/c5_step.py 3 bf16 > /dev/null 2>&1
