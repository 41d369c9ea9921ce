export TMPDIR=/tmp
root=$PWD
export PYTHONPATH=$root
for m in one comp; do
  mkdir -p /tmp/tl_$m
  (cd /tmp && rocprofv3 --kernel-trace -d /tmp/tl_$m -o tl_$m -- python $root/tools/asg_onecall_timeline.py $m) > gpurun_out/r06_run56_$m.log 2>&1
  db=$(find /tmp/tl_$m -name "*.db" | head -1)
  python tools/asg_onecall_timeline.py dump $db > gpurun_out/r06_run56_timeline_$m.txt 2>&1
done
python -m pytest tests/test_gpu_asg_small.py tests/test_gpu_criterion.py tests/test_gpu_criterion_fuzz.py tests/test_gpu_fl_compat.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/r06_run56_tests.log
