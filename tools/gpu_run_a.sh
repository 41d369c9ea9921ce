#!/bin/bash
# one GPU-box session: parity tests, GEMM A/B (LDS-DMA persistent kernel vs register-staged), FCC stream A/B, bench
mkdir -p gpurun_out
tag=${1:-r6}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
for r in 1 2; do
  timeout 300 python tools/gpu_probe.py gemm > gpurun_out/${tag}_gemm_glds_$r.log 2>&1
  W2L_GEMM_GLDS=0 timeout 300 python tools/gpu_probe.py gemm > gpurun_out/${tag}_gemm_v1_$r.log 2>&1
done
grep "sk=1" gpurun_out/${tag}_gemm_glds_2.log; echo; grep "sk=1" gpurun_out/${tag}_gemm_v1_2.log
timeout 300 python tools/gpu_probe.py fccbig conv > gpurun_out/${tag}_fccbig_wpc2.log 2>&1
W2L_FCC_WPC=3 timeout 300 python tools/gpu_probe.py fccbig > gpurun_out/${tag}_fccbig_wpc3.log 2>&1
W2L_FCC_WPC=1 timeout 300 python tools/gpu_probe.py fccbig > gpurun_out/${tag}_fccbig_wpc1.log 2>&1
grep fccbig gpurun_out/${tag}_fccbig_wpc*.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json
