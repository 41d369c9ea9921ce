python -m pytest tests/test_gpu_asg_small.py -m gpu -x -q -k "one_call or asg or fac" 2>&1 | grep -E "passed|failed|rror" > gpurun_out/r06_run68_tests.log
for i in 1 2 3; do python tools/asg_leg.py 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['fwd_ms'], d['fwd_bwd_ms'], d['fcc_fwd_ms'], d['fac_fwd_ms'], d['composed_calls']['fwd_ms'], d['composed_calls']['fwd_bwd_ms'], d['composed_calls']['bit_identical_to_one_call'])"; done > gpurun_out/r06_run68_asg_leg.log 2>&1
export TMPDIR=/tmp
root=$PWD
export PYTHONPATH=$root
mkdir -p /tmp/tl_one
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/tl_one -o tl_one -- python $root/tools/asg_onecall_timeline.py one) > gpurun_out/r06_run68_one.log 2>&1
db=$(find /tmp/tl_one -name "*.db" | head -1)
python tools/asg_onecall_timeline.py dump $db > gpurun_out/r06_run68_timeline_one.txt 2>&1
