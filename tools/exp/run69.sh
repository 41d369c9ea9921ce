# final tree (one-launch ASG forward pass): the whole -m gpu suite, smoke, the evidence run
(time python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" ) > gpurun_out/r06_run69_gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r06_run69_smoke.log
bash tools/evidence.sh r06_run69 > gpurun_out/r06_run69_evidence.log 2>&1
