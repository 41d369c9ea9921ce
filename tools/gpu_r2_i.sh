#!/bin/bash
# run 15: MFMA issue micro-benchmarks, conv probes (rs3 / rsf3 vs the previous generation, ablations, per-role cycle counts),
# rocprofv3 kernel statistics of the bench step, SQ / LDS counters of the conv kernels
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_rate tools/micro/mfma_rate.hip && /tmp/mfma_rate > gpurun_out/r2i_mfma_rate.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_valu tools/micro/mfma_valu.hip && /tmp/mfma_valu > gpurun_out/r2i_mfma_valu.log 2>&1
tail -3 gpurun_out/r2i_mfma_rate.log
timeout 300 python tools/conv_rs3.py --abl > gpurun_out/r2i_conv_rs3.log 2>&1; tail -5 gpurun_out/r2i_conv_rs3.log
timeout 200 python tools/conv_rsf3_scale.py > gpurun_out/r2i_conv_rsf3_scale.log 2>&1; tail -2 gpurun_out/r2i_conv_rsf3_scale.log
ARGS="bench.py --steps 5 --warmup 2 --no-asg --no-stress --no-c4 --no-c3 --no-cpu-baseline"
bash tools/prof.sh r2i_bench $ARGS; echo "prof rc=$?"
head -16 gpurun_out/r2i_bench_kernel_stats.csv | cut -c1-180
for c in 10 18; do
  bash tools/pmc.sh r2i_rs3_c${c}_a "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/conv_one3.py $c
  bash tools/pmc.sh r2i_rsf3_c${c}_a "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/conv_one3f.py $c
  grep -h "rs3\|rsf\|^kernel" gpurun_out/r2i_rs3_c${c}_a_pmc.csv gpurun_out/r2i_rsf3_c${c}_a_pmc.csv | cut -c1-300
done
