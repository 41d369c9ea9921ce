export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so
for m in 0 1 2; do
  echo "== W2L_GEMM_PRIO=$m"
  W2L_GEMM_PRIO=$m python tools/gemm_wg_times.py 2>&1 | grep "per CU"
done > gpurun_out/r06_run36_gemm_priority_modes.log 2>&1
for rep in 1 2; do for m in 0 1; do
  W2L_GEMM_PRIO=$m python tools/gemm_step_shapes.py prio=$m 2>&1 | tail -8
done; done >> gpurun_out/r06_run36_gemm_priority_modes.log 2>&1
