export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so W2L_GEMM_T160=0 W2L_GEMM_BUF=0
for a in 0 2 4 32 16 1 3 7 8 9 15; do
  W2L_GEMM_ABL=$a python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | grep -v done | sed "s/^/[global-address kernel abl=$a] /"
done > gpurun_out/r06_run43_gemm128g_ablations_global_address_kernel.log 2>&1
