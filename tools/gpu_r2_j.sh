#!/bin/bash
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PYTHONPATH TMPDIR=/tmp
mkdir -p /tmp/asgtl
root=$PWD; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/asgtl -o asg -- python $root/tools/asg_timeline.py) > gpurun_out/r2j_run.log 2>&1
f=$(find /tmp/asgtl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last iteration: take the last 40 kernels
t0 = None
sel = rows[-34:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} -> {e/1e3:9.1f} us  ({(e-s)/1e3:7.1f})  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:80]}")
P
