#!/bin/bash
# GPU session M: parity, conv probe, bench
mkdir -p gpurun_out
tag=${1:-r19}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log | cut -c1-200
for a in 0 6 0; do
  W2L_TDS_ABL=$a timeout 300 python tools/gpu_probe.py conv 2>&1 | grep "conv\] tds" | sed "s/^/[abl=$a] /" | tee -a gpurun_out/${tag}_conv.log
done
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['tds_conv'], d['asg_stress']['roofline']['achieved'])"
