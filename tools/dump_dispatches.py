"""per-dispatch rows of a rocprofv3 rocpd kernel trace for kernels whose name contains a pattern: start (us since the first
dispatch), duration (us), grid / workgroup sizes.   usage: dump_dispatches.py <results.db> <pattern> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
want = [c for c in cols if any(k in c.lower() for k in ("grid", "workgroup", "lds", "scratch"))]
q = "select name, start, end" + "".join(", " + c for c in want) + " from kernels order by start"
rows = list(db.execute(q))
t0 = rows[0][1] if rows else 0
out = ["kernel,start_us,dur_us," + ",".join(want)]
for r in rows:
    if sys.argv[2] in r[0]:
        out.append('"%s",%.1f,%.1f,%s' % (r[0][:60].replace('"', "'"), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, ",".join(str(v) for v in r[3:])))
text = "\n".join(out) + "\n"
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text)
else:
    sys.stdout.write(text)
