#!/bin/bash
# GPU session K: full parity with the LDS-DMA FCC ring as default, ring variants, conv, bench + rocprof stats + PMC traffic
mkdir -p gpurun_out
tag=${1:-r16}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log | cut -c1-200
: > gpurun_out/${tag}_fcc.log
for v in 1 2 3 4 0 1 3; do
  W2L_FCC_DMA=$v timeout 300 python tools/gpu_probe.py fccstream 2>&1 | grep "launches" | sed "s/^/[dma=$v] /" >> gpurun_out/${tag}_fcc.log
done
cat gpurun_out/${tag}_fcc.log
timeout 300 python tools/gpu_probe.py conv 2>&1 | grep "conv\] tds" | tee gpurun_out/${tag}_conv.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json | cut -c1-600
W2L_FCC_DMA=3 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg > gpurun_out/${tag}_bench_dma3.json 2>> gpurun_out/${tag}_bench.err
python -c "
import json
for f in ('${tag}_bench','${tag}_bench_dma3'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['asg_stress']['roofline']['achieved'], d['asg_stress']['roofline']['avg_launch_us'], d['asg_stress']['fwd_ms'], d['asg_stress']['bwd_ms'])"
