#!/bin/bash
# GPU session C: parity (wide epilogue, filter2, FCC RT=4, bucket events), A/B of each, bench
mkdir -p gpurun_out
tag=${1:-r8}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -15 gpurun_out/${tag}_tests.log
: > gpurun_out/${tag}_gemm_abl.log
for abl in 0 8 16 32 1; do
  W2L_GEMM_ABL=$abl timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd >> gpurun_out/${tag}_gemm_abl.log
done
W2L_GEMM_WIDE=0 timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | sed 's/^/[wide=0] /' >> gpurun_out/${tag}_gemm_abl.log
cat gpurun_out/${tag}_gemm_abl.log
timeout 300 python tools/gpu_probe.py gemm > gpurun_out/${tag}_gemm.log 2>&1; grep "sk=1" gpurun_out/${tag}_gemm.log
timeout 300 python tools/gpu_probe.py conv fccbig > gpurun_out/${tag}_conv_fcc.log 2>&1
W2L_TDS_FILTER_V1=1 W2L_FCC_RT=2 timeout 300 python tools/gpu_probe.py conv fccbig > gpurun_out/${tag}_conv_fcc_old.log 2>&1
grep -h "conv\|fccbig" gpurun_out/${tag}_conv_fcc.log gpurun_out/${tag}_conv_fcc_old.log
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-asg > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asg --no-stress --force-dist > gpurun_out/${tag}_bench_dist1.json 2> gpurun_out/${tag}_bench_dist1.err; echo "bench force-dist rc=$?"
cat gpurun_out/${tag}_bench_dist1.json | cut -c1-300; tail -3 gpurun_out/${tag}_bench_dist1.err
