python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/r06_run46_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_run46_smoke.log 2>&1
python bench.py > gpurun_out/r06_run46_bench.json 2> gpurun_out/r06_run46_bench.err
