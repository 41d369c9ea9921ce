# one-call ASG (w2l_asg_forward / w2l_asg_backward = the fused sequence): parity, then the ASG leg through it
python -m pytest tests/test_gpu_asg_small.py -m gpu -x -q -k "one_call" 2>&1 | tail -15 > gpurun_out/r06_run55_tests.log
for i in 1 2 3; do python tools/asg_leg.py 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['fwd_ms'], d['fwd_bwd_ms'], d['fcc_fwd_ms'], d['fac_fwd_ms'], d['composed_calls'])"; done > gpurun_out/r06_run55_asg_leg.log 2>&1
python -m pytest tests/test_gpu_asg_small.py tests/test_gpu_criterion.py tests/test_gpu_criterion_fuzz.py tests/test_gpu_fl_compat.py tests/test_gpu_pipeline.py tests/test_gpu_trainer.py -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/r06_run55_tests.log
