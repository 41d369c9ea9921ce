#!/bin/bash
# GPU session Q: parity, ASG criterion timing with FCC || FAC
mkdir -p gpurun_out
tag=${1:-r25}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log | cut -c1-200
timeout 300 python tools/gpu_probe.py asg 2>&1 | grep "asg" | tee gpurun_out/${tag}_asg.log
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stress > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json
d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['value'], d['ms_per_step']); print(d['asg_loss_ms_per_step'])"
