#!/bin/bash
# GPU session D: parity (batched conv staging), conv probe, FCC stream A/B + ablations + counters, GEMM tile-overhead ablations
mkdir -p gpurun_out
tag=${1:-r9}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log
timeout 300 python tools/gpu_probe.py conv 2>&1 | grep conv | tee gpurun_out/${tag}_conv.log
: > gpurun_out/${tag}_fcc.log
for v in "" "W2L_FCC_RING=1" "W2L_FCC_RT=4" "W2L_FCC_ABL=1" "W2L_FCC_ABL=2" "W2L_FCC_ABL=3" "W2L_FCC_WPC=1" ""; do
  env $v timeout 300 python tools/gpu_probe.py fccstream 2>&1 | grep fccstream >> gpurun_out/${tag}_fcc.log
done
cat gpurun_out/${tag}_fcc.log
: > gpurun_out/${tag}_gemm_abl.log
for abl in 0 9 1 8; do
  W2L_GEMM_ABL=$abl timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd >> gpurun_out/${tag}_gemm_abl.log
done
W2L_GEMM_SK=0 timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | sed 's/^/[sk=0] /' >> gpurun_out/${tag}_gemm_abl.log
W2L_GEMM_SK=0 W2L_GEMM_ABL=9 timeout 120 python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | sed 's/^/[sk=0] /' >> gpurun_out/${tag}_gemm_abl.log
cat gpurun_out/${tag}_gemm_abl.log
(cd /tmp && rocprofv3 -L > $OLDPWD/gpurun_out/${tag}_counters_list.txt 2>&1)
wc -l gpurun_out/${tag}_counters_list.txt
timeout 600 bash tools/pmc.sh ${tag}_fcc_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" tools/fcc_one.py
timeout 600 bash tools/pmc.sh ${tag}_fcc_tcc "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" tools/fcc_one.py
timeout 600 bash tools/pmc.sh ${tag}_fcc_fetch "FETCH_SIZE" tools/fcc_one.py
grep "fcc_big_gemm\|kernel," gpurun_out/${tag}_fcc_sq_pmc.csv gpurun_out/${tag}_fcc_tcc_pmc.csv gpurun_out/${tag}_fcc_fetch_pmc.csv | cut -c1-300
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-asg > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
cat gpurun_out/${tag}_bench.json | cut -c1-1500
