python -m pytest tests/test_gpu_nn.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/r06_run27_tests.log
python tools/gemm_epilogue_cost.py > gpurun_out/r06_run27_epilogue_cost.log 2>&1
F="--no-asg --no-stress --no-c3 --no-c4 --no-c5 --no-cpu-baseline --no-train-binary --no-input-pipeline"
for rep in 1 2; do
  python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused', d['value'], d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  W2L_NET_NOFUSE=1 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('plain', d['value'], d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done > gpurun_out/r06_run27_headline_fuse_ab.log 2>&1
for rep in 1 2; do
  for m in bf16 f32; do
    echo "fused $m: $(python tools/c3_step.py 5 $m 2>&1 | grep '\[c3\]' | cut -c1-230)"
    echo "plain $m: $(W2L_NET_NOFUSE=1 python tools/c3_step.py 5 $m 2>&1 | grep '\[c3\]' | cut -c1-230)"
  done
done > gpurun_out/r06_run27_c3_fuse_ab.log 2>&1
