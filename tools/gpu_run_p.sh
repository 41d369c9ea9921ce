#!/bin/bash
# GPU session P: parity (loader-wave GEMM), GEMM A/B loader vs 4-wave, bench both
mkdir -p gpurun_out
tag=${1:-r24}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -6 gpurun_out/${tag}_tests.log | cut -c1-300
for r in 1 2; do
  W2L_GEMM_LOADER=1 timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | sed 's/^/[loader=1] /' | tee -a gpurun_out/${tag}_gemm.log
  timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | sed 's/^/[loader=0] /' | tee -a gpurun_out/${tag}_gemm.log
done
W2L_GEMM_LOADER=1 timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-asg --no-stress > gpurun_out/${tag}_bench_loader.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-asg --no-stress > gpurun_out/${tag}_bench.json 2>> gpurun_out/${tag}_bench.err
python -c "
import json
for f in ('${tag}_bench_loader','${tag}_bench'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"
