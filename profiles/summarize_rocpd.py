"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the per-kernel
stats table that `--stats` prints: calls, total ms, average us, share of GPU kernel time.
usage: python profiles/summarize_rocpd.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                           "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,percent"]
    for n, c, s, a, mn, mx in rows:
        lines.append('"%s",%d,%.3f,%.1f,%.1f,%.1f,%.2f' % (n.replace('"', "'"), c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3,
                                                            100.0 * s / tot))
    lines.append('"TOTAL",%d,%.3f,,,,100.00' % (sum(r[1] for r in rows), tot / 1e6))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
