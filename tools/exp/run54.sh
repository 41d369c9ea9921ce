# fused ASG launch sequence (criterion_asg_fused.hpp): parity tests, then the ASG leg
python -m pytest tests/test_gpu_asg_small.py tests/test_gpu_criterion.py tests/test_gpu_criterion_fuzz.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_run54_tests.log
for i in 1 2 3; do python tools/asg_leg.py 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['fwd_ms'], d['fwd_bwd_ms'], d['fcc_fwd_ms'], d['fac_fwd_ms'])"; done > gpurun_out/r06_run54_asg_leg.log 2>&1
python -m pytest tests/test_gpu_fl_compat.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/r06_run54_tests.log
