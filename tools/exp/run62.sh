# final tree: the whole -m gpu suite, smoke, the default bench line
(time python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" ) > gpurun_out/r06_run62_gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r06_run62_smoke.log
(time timeout 900 python bench.py) > gpurun_out/r06_run62_bench.log 2> gpurun_out/r06_run62_bench.err
python tools/check_evidence.py gpurun_out/r06_run62_bench.log >> gpurun_out/r06_run62_smoke.log 2>&1
