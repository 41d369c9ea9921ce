#!/bin/bash
# GPU session F: parity (persistent conv fwd, buffer LDS-DMA default), conv A/B, bench, PMC traffic of the bench step
mkdir -p gpurun_out
tag=${1:-r11}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -12 gpurun_out/${tag}_tests.log
timeout 300 python tools/gpu_probe.py conv fccstream 2>&1 | grep "conv\|fccstream" | tee gpurun_out/${tag}_conv.log
W2L_TDS_FWD_V1=1 timeout 300 python tools/gpu_probe.py conv 2>&1 | grep "conv" | sed 's/^/[fwd v1] /' | tee -a gpurun_out/${tag}_conv.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json | cut -c1-1800
timeout 900 bash tools/pmc.sh ${tag}_fetch "FETCH_SIZE" bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg --stress-frames 40
timeout 900 bash tools/pmc.sh ${tag}_write "WRITE_SIZE" bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg --stress-frames 40
python tools/pmc_traffic.py gpurun_out/${tag}_fetch_pmc.csv gpurun_out/${tag}_write_pmc.csv gpurun_out/${tag}_pmc_traffic.json
timeout 900 bash tools/prof.sh ${tag}_bench bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asg --stress-frames 100
head -30 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-200
