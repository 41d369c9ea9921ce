export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so W2L_GEMM_T160=0
for a in 0 64 8 72 1 9; do
  W2L_GEMM_ABLBUF=$a python tools/gpu_probe.py gemmfwd 2>&1 | grep gemmfwd | grep -v done | sed "s/^/[ablbuf=$a] /"
done > gpurun_out/r06_run29_gemm128g_ablations.log 2>&1
