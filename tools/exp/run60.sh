# final tree: the whole -m gpu suite, smoke, then the evidence run (default bench line, per-leg kernel statistics, counter passes)
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -4) > gpurun_out/r06_run60_gpu_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_run60_smoke.log 2>&1
bash tools/evidence.sh r06_run60 > gpurun_out/r06_run60_evidence.log 2>&1
