export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so
for rep in 1 2; do
for st in 0 50 30 70 200; do
  W2L_GEMM_STAGGER=$st python tools/gemm_step_shapes.py stagger=$st 2>&1 | tail -8
done
done > gpurun_out/r06_run19_gemm_stagger.log 2>&1
tail -3 gpurun_out/r06_run19_gemm_stagger.log
