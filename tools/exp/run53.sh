# A-direct K loop of the 128 x 160 kernel: parity tests, then the step's GEMM shapes A/B (probe library, interleaved)
python -m pytest tests/test_gpu_nn.py -m gpu -x -q -k "gemm or linear" 2>&1 | tail -3 > gpurun_out/r06_run53_tests.log
export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so
for r in 1 2; do
  W2L_GEMM_ADIR=0 python tools/gemm_step_shapes.py lds 2>&1 | grep "^\["
  W2L_GEMM_ADIR=1 python tools/gemm_step_shapes.py adir 2>&1 | grep "^\["
  W2L_GEMM_ADIR=1 W2L_GEMM_T160=2 python tools/gemm_step_shapes.py adir_160everywhere 2>&1 | grep "^\["
done > gpurun_out/r06_run53_gemm_adir_ab.log 2>&1
