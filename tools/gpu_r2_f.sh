#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
timeout 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_trainer.py -m gpu -q -x -k "bf16 or mixed_precision" > gpurun_out/r2f_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r2f_tests.log | cut -c1-400
timeout 400 python bench.py --no-stress --no-c4 --no-asg --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; echo "bench rc=$?"
grep "^\[bench" gpurun_out/r2f_bench.err; python - <<'P'
import json
d=json.load(open('gpurun_out/r2f_bench.json'))
print(json.dumps(d.get('streaming_tds_bf16_step'))[:1500]); print(json.dumps(d['roofline']['tds_conv'])[:700])
P
