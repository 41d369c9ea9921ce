export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so
for rep in 1 2; do for m in 0 1; do W2L_LN_WAVE=$m python tools/ln_small_one.py 2>&1 | grep wave=; done; done > gpurun_out/r06_run37_ln_wave_per_group.log
unset W2L_HIP_SO
python -m pytest tests/test_gpu_nn.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/r06_run37_tests.log
for rep in 1 2; do
  echo "wave bf16: $(W2L_USE_PROBE=1 W2L_LN_WAVE=1 python tools/c3_step.py 5 bf16 2>&1 | grep '\[c3\]' | cut -c1-230)"
  echo "block bf16: $(W2L_USE_PROBE=1 W2L_LN_WAVE=0 python tools/c3_step.py 5 bf16 2>&1 | grep '\[c3\]' | cut -c1-230)"
done > gpurun_out/r06_run37_c3_ln_ab.log 2>&1
