#!/bin/bash
# full -m gpu suite + default bench on the current tree (run 14)
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2g_tests.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2g_tests.log | cut -c1-400
timeout 600 python bench.py > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err; echo "bench rc=$?"
grep "^\[bench" gpurun_out/r2g_bench.err | tail -30
python - <<'P'
import json
d=json.load(open('gpurun_out/r2g_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','cpu_baseline')})
print(json.dumps(d['roofline'])[:1800])
P
