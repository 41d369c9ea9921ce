#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <python args...>  -> gpurun_out/<tag>_pmc.csv  (counter pass only: no tracing flags)
tag=$1; shift; ctr=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
root=$PWD
script=$root/$1; shift
export PYTHONPATH=$root:$PYTHONPATH
mkdir -p $out /tmp/pmc_$tag
(cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o $tag -- python $script "$@") > $out/${tag}_pmc_run.log 2>&1
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python - "$f" > $out/${tag}_pmc.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    k = r["Kernel_Name"][:90]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel,dispatches," + ",".join(names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print('"%s",%d,' % (k, cnt[k]) + ",".join("%.4g" % (v.get(n, 0) / max(1, cnt[k])) for n in names))
PY
fi
ls /tmp/pmc_$tag/* >> $out/${tag}_pmc_run.log 2>&1
