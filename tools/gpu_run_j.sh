#!/bin/bash
# GPU session J: FCC LDS-DMA ring kernel (parity + timing), conv K-loop ablation without A reads
mkdir -p gpurun_out
tag=${1:-r15}
W2L_FCC_DMA=1 timeout 900 python -m pytest tests/test_gpu_criterion.py -m gpu -x -q -k "large_n or north_star" > gpurun_out/${tag}_tests_dma.log 2>&1; echo "pytest dma rc=$?" >> gpurun_out/${tag}_tests_dma.log
tail -4 gpurun_out/${tag}_tests_dma.log | cut -c1-200
: > gpurun_out/${tag}_fcc.log
for v in "" "W2L_FCC_DMA=1" "" "W2L_FCC_DMA=1"; do
  env $v timeout 300 python tools/gpu_probe.py fccstream 2>&1 | grep fccstream | sed "s/^/[$v] /" >> gpurun_out/${tag}_fcc.log
done
cat gpurun_out/${tag}_fcc.log
for a in 0 6 14; do
  W2L_TDS_ABL=$a timeout 300 python tools/gpu_probe.py conv 2>&1 | grep "conv\] tds" | sed "s/^/[abl=$a] /" | tee -a gpurun_out/${tag}_conv.log
done
W2L_FCC_DMA=1 timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asg > gpurun_out/${tag}_bench_dma.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench_dma.json')); print(d['asg_stress'])"
