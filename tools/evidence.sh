#!/bin/bash
# evidence run on the final tree (rounds 3 - 5): the default bench line, rocprofv3 kernel statistics per bench leg, PMC traffic passes
# usage (on the GPU box): bash tools/evidence.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-ev}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 900 python bench.py) > gpurun_out/${tag}_bench.log 2> gpurun_out/${tag}_bench.err
common="--no-cpu-baseline --no-input-pipeline --no-oracle-checks --no-train-binary"
timeout 400 bash tools/prof.sh ${tag}_headline_asg_ctc bench.py --steps 3 --warmup 1 --no-stress --no-c4 --no-c3 --no-c5 $common
timeout 400 bash tools/prof.sh ${tag}_stress bench.py --steps 1 --warmup 0 --no-asg --no-c4 --no-c3 --no-c5 $common
timeout 400 bash tools/prof.sh ${tag}_c4 bench.py --steps 1 --warmup 0 --no-asg --no-stress --no-c3 --no-c5 $common
timeout 300 bash tools/prof.sh ${tag}_c3 tools/c3_step.py 3 bf16
timeout 300 bash tools/prof.sh ${tag}_c5 tools/c5_step.py 3 bf16
# HBM-side traffic (separate counter passes, no tracing domains): headline GEMM + the alpha-pass stream + the bf16 legs
# (--no-c3: the config-3 fp32 leg launches the same GEMM kernels as the headline -- with it the "gemm_lds_dma" group mixed the two
#  workloads and roofline.traffic was not the headline kernel's own bytes: round-4 verdict, evidence hygiene)
timeout 500 bash tools/pmc.sh ${tag}_fetch "FETCH_SIZE" bench.py --steps 1 --warmup 0 --no-asg --no-c3 --no-c4 --no-c5 --stress-frames 40 $common
timeout 500 bash tools/pmc.sh ${tag}_write "WRITE_SIZE" bench.py --steps 1 --warmup 0 --no-asg --no-c3 --no-c4 --no-c5 --stress-frames 40 $common
# the RCCL path of the headline step with one rank (process group, bucketed reducer on the side stream, batch size through the arena's tail)
(timeout 300 python bench.py --force-dist --steps 4 --warmup 2 --no-asg --no-stress --no-c4 --no-c3 --no-c5 $common) > gpurun_out/${tag}_bench_force_dist_1rank.log 2> gpurun_out/${tag}_bench_force_dist_1rank.err
python tools/pmc_traffic.py gpurun_out/${tag}_fetch_pmc.csv gpurun_out/${tag}_write_pmc.csv gpurun_out/${tag}_pmc_traffic.json > /dev/null 2>&1

# every record kept from this run must parse (a stdout without its bench line is not evidence)
python tools/check_evidence.py gpurun_out/${tag}_bench.log gpurun_out/${tag}_bench_force_dist_1rank.log gpurun_out/${tag}_pmc_traffic.json \
  gpurun_out/${tag}_headline_asg_ctc_kernel_stats.csv gpurun_out/${tag}_stress_kernel_stats.csv gpurun_out/${tag}_c4_kernel_stats.csv \
  gpurun_out/${tag}_c3_kernel_stats.csv gpurun_out/${tag}_c5_kernel_stats.csv || { echo "evidence run INCOMPLETE"; exit 1; }
echo done
