#!/bin/bash
# final GPU session of a round: full parity suite, full bench (cpu_baseline + ASG + stress + C4 legs), rocprofv3 kernel
# statistics and the FETCH_SIZE / WRITE_SIZE counter passes of the same bench command
mkdir -p gpurun_out
tag=${1:-rfinal}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
grep -E "passed|failed|rror|pytest rc" gpurun_out/${tag}_tests.log | tail -4 | cut -c1-200
timeout 900 bash tools/pmc.sh ${tag}_fetch "FETCH_SIZE" bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg --no-c4 --stress-frames 40
timeout 900 bash tools/pmc.sh ${tag}_write "WRITE_SIZE" bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-asg --no-c4 --stress-frames 40
python tools/pmc_traffic.py gpurun_out/${tag}_fetch_pmc.csv gpurun_out/${tag}_write_pmc.csv gpurun_out/${tag}_pmc_traffic.json | cut -c1-600
cp gpurun_out/${tag}_pmc_traffic.json profiles/r01_${tag}_pmc_traffic.json
timeout 900 bash tools/prof.sh ${tag}_bench bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-asg --no-c4 --stress-frames 100
head -14 gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-170
timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/${tag}_bench.json"))
r=d["roofline"]; print("utt/s", d["value"], "ms", d["ms_per_step"], "gemm TF", r["achieved"], r["frac"], "traffic", r["traffic"])
a=d["asg_stress"]; print("asg stress fwd/bwd ms", a["fwd_ms"], a["bwd_ms"], "hbm frac", a["roofline"]["frac"], "traffic", a["roofline"]["traffic"])
print("asg loss", d["asg_loss_ms_per_step"]["fwd_ms"], d["asg_loss_ms_per_step"]["fwd_bwd_ms"])
print("c4", d.get("conv_glu_asg_step"))
print("cpu", d.get("cpu_baseline"))
PY
