# sanity of the tree at session start: full GPU suite, default bench line, T160-everywhere A/B on the step's GEMM shapes
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06_run20_gpu_tests.log
python bench.py > gpurun_out/r06_run20_bench.json 2> gpurun_out/r06_run20_bench.err
export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so
for rep in 1 2; do
for m in 1 2; do
  W2L_GEMM_T160=$m python tools/gemm_step_shapes.py t160=$m 2>&1 | tail -8
done
done > gpurun_out/r06_run20_gemm_t160_modes.log 2>&1
