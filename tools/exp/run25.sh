F="--no-asg --no-stress --no-c3 --no-c4 --no-c5 --no-cpu-baseline --no-train-binary --no-input-pipeline"
for rep in 1 2; do
  python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused', d['value'], d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
  W2L_NET_NOFUSE=1 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('plain', d['value'], d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done > gpurun_out/r06_run25_headline_fuse_ab.log 2>&1
