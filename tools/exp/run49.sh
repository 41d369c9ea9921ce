python -m pytest tests/test_gpu_asg_small.py tests/test_gpu_criterion.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/r06_run49_tests.log
for i in 1 2 3; do python tools/asg_leg.py 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print({k:d[k] for k in ('fwd_ms','fwd_bwd_ms','fcc_fwd_ms','fac_fwd_ms')})"; done > gpurun_out/r06_run49_asg_leg.log
bash tools/prof.sh r06_run49_asg tools/asg_leg.py > /dev/null 2>&1
