#!/bin/bash
# full -m gpu suite + smoke + default bench (driver-style flags) on the current tree (run 23)
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2t_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2t_tests.log | cut -c1-300
t1=$(date +%s); echo "tests took $((t1-t0)) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t2=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2t_bench.json 2> gpurun_out/r2t_bench.err; echo "bench rc=$?"
t3=$(date +%s); echo "bench took $((t3-t2)) s"
grep "^\[bench" gpurun_out/r2t_bench.err | tail -12
python - <<'P'
import json
d=json.load(open('gpurun_out/r2t_bench.json'))
print({k:d[k] for k in ('value','ms_per_step','cpu_baseline')})
print(json.dumps(d.get('transformer_ctc_step'))[:900])
print(json.dumps(d.get('asg_stress'))[:400])
print(json.dumps(d['roofline'])[:1200])
P
