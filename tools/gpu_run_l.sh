#!/bin/bash
# GPU session L: parity with the nontemporal FCC ring default; FCC 3 vs 4 in bench; conv counters
mkdir -p gpurun_out
tag=${1:-r17}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log | cut -c1-200
timeout 300 python tools/gpu_probe.py conv 2>&1 | grep "conv\] tds" | tee gpurun_out/${tag}_conv.log
timeout 900 python bench.py --steps 8 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
W2L_FCC_DMA=3 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg > gpurun_out/${tag}_bench_dma3.json 2>> gpurun_out/${tag}_bench.err
python -c "
import json
for f in ('${tag}_bench','${tag}_bench_dma3'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['asg_stress']['roofline']['achieved'], d['asg_stress']['roofline']['avg_launch_us'], d['asg_stress']['fwd_ms'], d['asg_stress']['bwd_ms'])"
timeout 600 bash tools/pmc.sh ${tag}_conv_sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" tools/conv_one.py
timeout 600 bash tools/pmc.sh ${tag}_conv_sq2 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" tools/conv_one.py
timeout 600 bash tools/pmc.sh ${tag}_conv_sq3 "SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU" tools/conv_one.py
grep "fwd2\|kernel," gpurun_out/${tag}_conv_sq*_pmc.csv | cut -c1-400
# PMC traffic of the bench step with the final kernels
timeout 900 bash tools/pmc.sh ${tag}_fetch "FETCH_SIZE" bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg --stress-frames 40
timeout 900 bash tools/pmc.sh ${tag}_write "WRITE_SIZE" bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg --stress-frames 40
python tools/pmc_traffic.py gpurun_out/${tag}_fetch_pmc.csv gpurun_out/${tag}_write_pmc.csv gpurun_out/${tag}_pmc_traffic.json
timeout 900 bash tools/prof.sh ${tag}_bench bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asg --stress-frames 100
head -12 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-160
