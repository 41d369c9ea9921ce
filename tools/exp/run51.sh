python -m pytest tests/test_gpu_asg_small.py tests/test_gpu_criterion.py -m gpu -x -q 2>&1 | tail -1 > gpurun_out/r06_run51_tests.log
export W2L_HIP_SO=$PWD/wav2letter_amd/libw2l_hip_probe.so
for i in 1 2 3; do
  for m in excl share; do
    if [ $m = share ]; then export W2L_ASG_SHARE_CUS=1; else unset W2L_ASG_SHARE_CUS; fi
    python tools/asg_leg.py 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$m', d['fwd_ms'], d['fwd_bwd_ms'], d['fcc_fwd_ms'], d['fac_fwd_ms'])"
  done
done > gpurun_out/r06_run51_asg_exclusive_cus.log
