#!/bin/bash
# round 2, call E: full GPU suite + default bench (headline, legs, host baseline) after the conv work
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r2e_tests.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2e_tests.log | cut -c1-300
timeout 400 python bench.py > gpurun_out/r2e_bench.json 2> gpurun_out/r2e_bench.err; echo "bench rc=$?"
grep "^\[bench" gpurun_out/r2e_bench.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r2e_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus')})
print(json.dumps(d['roofline']['tds_conv'])[:900]); print(json.dumps(d['roofline']['whole_step'])[:300]); print(d['asg_loss_ms_per_step']); print(d['asg_stress'].get('loss_check')); print(d['cpu_baseline'])
P
