#!/bin/bash
# GPU session N: parity, GEMM probe (epilogue prefetch), bench
mkdir -p gpurun_out
tag=${1:-r22}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log | cut -c1-200
timeout 300 python tools/gpu_probe.py gemm 2>&1 | grep "sk=1" | tee gpurun_out/${tag}_gemm.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'], d['roofline']['tds_conv']['ms_per_step'], d['asg_stress']['roofline']['achieved'])"
