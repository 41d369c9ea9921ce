#!/bin/bash
# usage: tools/prof.sh <tag> <python args...>   -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 kernel trace)
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
root=$PWD
script=$root/$1; shift
export PYTHONPATH=$root:$PYTHONPATH
mkdir -p $out /tmp/prof_$tag
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag -- python $script "$@") > $out/${tag}_run.log 2>&1
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
if [ -n "$db" ]; then python $out/../profiles/summarize_rocpd.py $db $out/${tag}_kernel_stats.csv; fi
csv=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -n "$csv" ]; then cp $csv $out/${tag}_rocprof_stats.csv; fi
ls /tmp/prof_$tag/* | head >> $out/${tag}_run.log
