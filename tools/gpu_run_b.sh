#!/bin/bash
# GPU session B: parity of the changed kernels, GEMM ablation ladder, FCC stream ring A/B, SQ counters of the LDS-DMA GEMM
mkdir -p gpurun_out
tag=${1:-r7}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -3 gpurun_out/${tag}_tests.log
: > gpurun_out/${tag}_gemm_abl.log
for abl in 0 1 2 3 4 7 8 15 0; do
  W2L_GEMM_ABL=$abl timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd >> gpurun_out/${tag}_gemm_abl.log
done
W2L_GEMM_GLDS=0 timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd >> gpurun_out/${tag}_gemm_abl.log
cat gpurun_out/${tag}_gemm_abl.log
timeout 300 python tools/gpu_probe.py fccbig > gpurun_out/${tag}_fccbig_wpc2.log 2>&1
W2L_FCC_WPC=1 timeout 300 python tools/gpu_probe.py fccbig > gpurun_out/${tag}_fccbig_wpc1.log 2>&1
grep -h fccbig gpurun_out/${tag}_fccbig_wpc*.log
timeout 600 bash tools/pmc.sh ${tag}_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" tools/gemm_one.py 4096 4096 4096 fwd
timeout 600 bash tools/pmc.sh ${tag}_sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" tools/gemm_one.py 4096 4096 4096 fwd
cat gpurun_out/${tag}_sq1_pmc.csv gpurun_out/${tag}_sq2_pmc.csv | cut -c1-400
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-asg > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json
