export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so
for rep in 1 2; do for m in 0 1; do
  W2L_GEMM_NTSTORE=$m python tools/gemm_step_shapes.py ntstore=$m 2>&1 | tail -8
done; done > gpurun_out/r06_run44_gemm_nontemporal_stores.log 2>&1
