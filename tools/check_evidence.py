#!/usr/bin/env python
"""Evidence hygiene: every file handed over as a record must parse.

  python tools/check_evidence.py <file> ...     (default: every *.json under profiles/)

A .json file must load as JSON, or -- a kept stdout of bench.py / a tools/ script -- hold a line that loads as a JSON object
(the record itself, possibly behind a "[tag] " prefix or next to library banners).  A *_kernel_stats.csv must have a header and at least one kernel row.  Exit code 1 names what does not.
(Round-5 verdict: profiles/r05_run28_bench_force_dist_1rank.json was 76 bytes of RCCL banner and no bench line.)"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(path):
    try:
        txt = open(path).read()
    except OSError as e:
        return f"unreadable ({e})"
    if path.endswith(".csv"):
        rows = [l for l in txt.splitlines() if l.strip()]
        return None if len(rows) >= 2 else "no kernel rows"
    try:
        json.loads(txt)
        return None
    except ValueError:
        pass
    lines = [l for l in txt.splitlines() if l.strip()]
    if not lines:
        return "empty"
    for l in lines:   # a kept stdout: some line (after a "[tag] " prefix, if any) is a JSON object -- the record itself
        l = l.strip()
        if l.startswith("[") and "] {" in l:
            l = l[l.index("] {") + 2:]
        if l.startswith("{"):
            try:
                if isinstance(json.loads(l), dict):
                    return None
            except ValueError:
                pass
    return "neither a JSON document nor a stdout with a JSON record line"


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json")))
    bad = [(f, why) for f in files for why in [check(f)] if why]
    for f, why in bad:
        print(f"check_evidence: {f}: {why}", file=sys.stderr)
    print(f"check_evidence: {len(files) - len(bad)} / {len(files)} files parse")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
